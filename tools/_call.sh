set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 300 python -m pytest tests/test_gpu_vposer_service.py -q > gpurun_out/r3q/svc.log 2>&1; echo "rc=$?" >> gpurun_out/r3q/svc.log
tail -3 gpurun_out/r3q/svc.log
for sets in 16 8 4; do
MVFIT_VP_SETS=$sets timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants --prior vposer > gpurun_out/r3q/bench_vp_$sets.log 2>&1
MVFIT_VP_SETS=$sets timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants --prior vposer --sparse > gpurun_out/r3q/bench_vp_sparse_$sets.log 2>&1
done
python - <<'PY'
import json
for n in ('bench_vp_16','bench_vp_sparse_16','bench_vp_8','bench_vp_sparse_8','bench_vp_4','bench_vp_sparse_4'):
    try:
        l=[x for x in open('gpurun_out/r3q/%s.log'%n) if x.startswith('{')]
        d=json.loads(l[-1]); print(n, d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], 'us/round %.2f' % (1e3*d['ms_per_step']/d['closure_rounds_per_fit']), d['vertex_passes_last_fit'])
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/r3q/%s.log'%n).read()[-1500:])
PY
