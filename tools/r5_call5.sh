set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
for rp in 2 3; do
  for cfg in configs3 configs4; do
    timeout 300 python bench.py --config $cfg --resident-pass $rp --no-pmc --no-variants --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_${cfg}_rp$rp.json.log 2> $O/bench_${cfg}_rp$rp.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$O/bench_${cfg}_rp$rp.json.log') if l.startswith('{')][-1]); r = d['roofline']
    print('$cfg resident=$rp', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r.get('avg_launch_is'), r['frac'], 'span', r.get('round_span_us'), 'slowest', r.get('slowest_workgroup_us'), 'busy', r.get('workgroup_busy_us'), 'alone', r.get('alone_per_round_us'), d['vertex_passes_last_fit'])
except Exception as e:
    print('$cfg resident=$rp FAILED', e)
PY
  done
done
