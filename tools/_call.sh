set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_vposer_service.py tests/test_gpu_reuse.py tests/test_gpu_trajectory.py tests/test_gpu_lbfgs.py tests/test_gpu_async.py tests/test_gpu_demo.py tests/test_gpu_sequence.py tests/test_gpu_sdf_term.py tests/test_gpu_large_batch.py -q > gpurun_out/r3q/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3q/tests.log
tail -4 gpurun_out/r3q/tests.log
