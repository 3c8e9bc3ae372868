set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n
rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_closure.py tests/test_gpu_async.py tests/test_gpu_vposer_service.py tests/test_gpu_sdf_term.py tests/test_gpu_init_guess.py tests/test_gpu_gather.py -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants > $O/bench.log 2>&1; grep -o '"value": [0-9.]*' $O/bench.log | head -1
