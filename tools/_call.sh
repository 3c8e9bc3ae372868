set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 600 python -m pytest tests/test_gpu_large_batch.py -q -x > gpurun_out/r3a/large_batch.log 2>&1; echo "rc=$?" >> gpurun_out/r3a/large_batch.log
timeout 300 python tests/quick_timing.py 32 > gpurun_out/r3a/quick32.log 2>&1
timeout 300 python tests/phase_timing.py > gpurun_out/r3a/phase.log 2>&1
tail -5 gpurun_out/r3a/large_batch.log; cat gpurun_out/r3a/quick32.log | tail -12; tail -30 gpurun_out/r3a/phase.log
