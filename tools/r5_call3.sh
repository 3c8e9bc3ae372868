# round 5, GPU call 3: options migration regression (changed tests), 8-rank dry run, resident timeline, VPoser set count
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_closure.py tests/test_gpu_closure_helpers.py tests/test_gpu_large_batch.py tests/test_gpu_sdf.py tests/test_gpu_sdf_cull.py tests/test_gpu_sdf_term.py tests/test_gpu_vposer_service.py -q -x -s > $O/changed.log 2>&1; echo "rc=$?" >> $O/changed.log; grep -E "passed|failed|rc=|compact vs|Error" $O/changed.log | tail -6
timeout 900 python -m pytest tests/test_gpu_bench_8rank.py -q -x -s > $O/rank8.log 2>&1; echo "rc=$?" >> $O/rank8.log; grep -E "passed|failed|rc=|8 ranks" $O/rank8.log | tail -4
for B in 32 128; do MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 200 python tests/vp_resident_timeline.py $B > $O/timeline_$B.log 2>&1; tail -6 $O/timeline_$B.log; done
for sets in 0 8; do
  timeout 300 python bench.py --prior vposer --vposer-sets $sets --no-pmc --no-variants --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_vposer_sets$sets.json.log 2> $O/bench_vposer_sets$sets.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_vposer_sets$sets.json.log').read().strip().splitlines()[-1])
    r = d['roofline']
    print('vposer sets=$sets', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], d['vertex_passes_last_fit'], d['decoder_helpers_last_fit'])
except Exception as e:
    print('vposer sets=$sets FAILED', e)
PY
done
timeout 300 python bench.py --config demo --no-pmc --no-variants --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_demo.json.log 2> $O/bench_demo.err; python -c "
import json; d=json.loads(open('$O/bench_demo.json.log').read().strip().splitlines()[-1]); print('demo', d['value'], d['ms_per_step'], d['vertex_passes_last_fit'])"
