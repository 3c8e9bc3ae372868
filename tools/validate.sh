# Developer tool for `gpurun -- "bash tools/validate.sh"`: the standard validation sequence (GPU suite, smoke, default bench, VPoser bench).
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/validate; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
grep -E "passed|failed|rc=" $O/gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json.log 2> $O/bench.err; grep -o '"value": [0-9.]*' $O/bench.json.log | head -1
timeout 400 python bench.py --prior vposer --no-pmc --no-cpu-baseline --no-variants > $O/bench_vposer.json.log 2> $O/bench_vposer.err; grep -o '"value": [0-9.]*' $O/bench_vposer.json.log | head -1
