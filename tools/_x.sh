cd $GRAFT_REPO_ROOT
O=gpurun_out/refill; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large_batch.py -q -x -s -k "work_queue" > $O/tests.log 2>&1; tail -12 $O/tests.log
