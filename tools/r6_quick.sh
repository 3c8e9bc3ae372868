# gpurun driver (scratch): bit-identity of the resident forms + the bench lines that show the pass
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6quick}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_demo.py -q -x -s > $O/async_tests.log 2>&1; tail -3 $O/async_tests.log
for CFG in "--config configs3" "--config configs4" "--frames 32"; do
  N=$(echo $CFG | tr -d ' -'); timeout 400 python bench.py $CFG --steps 5 --warmup 1 --no-pmc --no-cpu-baseline --no-variants > $O/bench_$N.json.log 2> $O/bench_$N.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/bench*.json.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), r.get('alone_per_round_us'), d.get('passes'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
