set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s
rm -rf $O; mkdir -p $O
export MVFIT_VP_PIPE14=1
timeout 900 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_closure.py tests/test_gpu_async.py -q > $O/tests.log 2>&1; grep -E "passed|failed|Error|assert " $O/tests.log | tail -6
for r in 1 2; do PYTHONPATH=. timeout 300 python tests/report_vertex_pass.py > $O/vp_$r.log 2>&1; grep "^B " $O/vp_$r.log; done
unset MVFIT_VP_PIPE14
PYTHONPATH=. timeout 300 python tests/report_vertex_pass.py > $O/vp_8.log 2>&1; grep "^B " $O/vp_8.log
