set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vposer_service.py -q > $O/tests.log 2>&1; tail -25 $O/tests.log
