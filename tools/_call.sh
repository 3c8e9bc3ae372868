set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_reuse.py -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-pmc > $O/bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3m/bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['variants']['time_to_solution_opt_in']))
PY
