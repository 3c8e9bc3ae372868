set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_closure.py tests/test_gpu_async.py -q -x > $O/tests.log 2>&1; grep -E "passed|failed|Error|assert" $O/tests.log | tail -5
for r in 1 2; do PYTHONPATH=. timeout 300 python tests/report_vertex_pass.py > $O/vp_$r.log 2>&1; grep "^B " $O/vp_$r.log; done
