set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 300 python tests/quick_async.py 32 128 161 256 > gpurun_out/r3c/quick_async.log 2>&1
timeout 900 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_async.py tests/test_gpu_sharded_fit.py tests/test_gpu_trajectory.py -q > gpurun_out/r3c/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/tests.log
cat gpurun_out/r3c/quick_async.log; tail -30 gpurun_out/r3c/tests.log
