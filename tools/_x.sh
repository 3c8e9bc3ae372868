cd $GRAFT_REPO_ROOT
O=gpurun_out/refill; rm -rf $O; mkdir -p $O
for CFG in "--frames 32" "--frames 256" "--frames 256 --work-queue 0" "--config configs3" "--prior gmm"; do
  N=$(echo $CFG | tr -d ' -'); timeout 400 python bench.py $CFG --steps 5 --warmup 1 --no-pmc --no-cpu-baseline --no-variants > $O/bench_$N.json.log 2> $O/bench_$N.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/bench*.json.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), d.get('vertex_passes_last_fit'), d.get('invalid_reason'))
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json.log','.err')).read()[-600:])
PY
timeout 900 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_async.py -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log
