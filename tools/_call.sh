set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 120 python tests/report_vertex_pass.py > gpurun_out/r3l/vp.log 2>&1
timeout 900 python -m pytest tests/test_gpu_closure.py tests/test_gpu_large_batch.py tests/test_gpu_async.py tests/test_gpu_sharded_fit.py -q > gpurun_out/r3l/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3l/tests.log
timeout 120 python tests/quick_async.py 128 256 > gpurun_out/r3l/quick_async.log 2>&1
cat gpurun_out/r3l/vp.log; tail -25 gpurun_out/r3l/tests.log; cat gpurun_out/r3l/quick_async.log
