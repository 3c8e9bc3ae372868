# round 5, GPU call 1: the resident vertex pass (parity + timing) and the staged reference (real caller, cpu baseline)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py -q -x > $O/async.log 2>&1; echo "rc=$?" >> $O/async.log; tail -5 $O/async.log
timeout 400 python -m pytest tests/test_gpu_large_batch.py -q -x > $O/large.log 2>&1; echo "rc=$?" >> $O/large.log; tail -3 $O/large.log
for r in auto 0; do
  if [ $r = auto ]; then unset MVFIT_VP_RESIDENT; else export MVFIT_VP_RESIDENT=$r; fi
  for cfg in configs1 configs3; do
    timeout 300 python bench.py --config $cfg --no-pmc --no-variants --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_${cfg}_res${r}.json.log 2> $O/bench_${cfg}_res${r}.err
    python - <<PY
import json
try:
    d = json.loads(open('$O/bench_${cfg}_res${r}.json.log').read().strip().splitlines()[-1])
    r = d['roofline']
    print('$cfg resident=${r}', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'], r.get('alone_per_round_us'), r.get('workgroup_busy_us'), d['vertex_passes_last_fit'])
except Exception as e:
    print('$cfg resident=${r} FAILED', e)
PY
  done
done
unset MVFIT_VP_RESIDENT
timeout 900 python -m pytest tests/test_gpu_real_caller.py -q -x -s > $O/real_caller.log 2>&1; echo "rc=$?" >> $O/real_caller.log; grep -E "real caller:|passed|failed|rc=" $O/real_caller.log | tail -8
