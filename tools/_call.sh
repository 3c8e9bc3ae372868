set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout 600 python -m pytest tests/test_gpu_reuse.py tests/test_gpu_lbfgs.py tests/test_gpu_trajectory.py tests/test_gpu_async.py -q > gpurun_out/r3p/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3p/tests.log
tail -4 gpurun_out/r3p/tests.log
for rep in 1 2; do
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_head.so timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants > gpurun_out/r3p/bench_head$rep.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants > gpurun_out/r3p/bench_new$rep.log 2>&1
done
timeout 300 python bench.py --no-cpu-baseline --no-pmc > gpurun_out/r3p/bench.log 2>&1
python - <<'PY'
import json
for n in ('bench_head1','bench_new1','bench_head2','bench_new2','bench'):
    l=[x for x in open('gpurun_out/r3p/%s.log'%n) if x.startswith('{')]
    d=json.loads(l[-1]); print(n, d['value'], d['ms_per_step'])
print(json.dumps(d['variants']['time_to_solution_opt_in']))
PY
