set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py tests/test_gpu_large_batch.py tests/test_gpu_bench_8rank.py tests/test_gpu_demo.py -q -x -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|8 ranks|demo, 48|share outside" $O/tests.log | tail -8
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 200 python tests/vp_resident_timeline.py 128 > $O/timeline_128.log 2>&1; tail -6 $O/timeline_128.log
timeout 300 python bench.py --config configs3 --no-pmc --no-variants --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_configs3.json.log 2> $O/bench_configs3.err
timeout 600 python bench.py --no-variants --no-cpu-baseline > $O/bench.json.log 2> $O/bench.err
python - <<PY
import json
for f in ('bench_configs3', 'bench'):
    try:
        d = json.loads([l for l in open('$O/%s.json.log' % f) if l.startswith('{')][-1]); r = d['roofline']
        print(f, d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'], r.get('slowest_workgroup_us'), r.get('alone_per_round_us'), 'traffic', r.get('traffic'), r.get('mfma_util'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
