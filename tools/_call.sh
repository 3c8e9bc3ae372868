set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 300 python -m pytest tests/test_gpu_vposer_service.py -q > gpurun_out/r3q/svc.log 2>&1; echo "rc=$?" >> gpurun_out/r3q/svc.log
tail -30 gpurun_out/r3q/svc.log
timeout 600 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_lbfgs.py tests/test_gpu_demo.py tests/test_gpu_sequence.py -q > gpurun_out/r3q/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3q/tests.log
tail -5 gpurun_out/r3q/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants --prior vposer > gpurun_out/r3q/bench_vp.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants --prior vposer --sparse > gpurun_out/r3q/bench_vp_sparse.log 2>&1
python - <<'PY'
import json
for n in ('bench_vp','bench_vp_sparse'):
    try:
        l=[x for x in open('gpurun_out/r3q/%s.log'%n) if x.startswith('{')]
        d=json.loads(l[-1]); print(n, d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['vertex_passes_last_fit'])
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/r3q/%s.log'%n).read()[-1500:])
PY
cp mvsmplfitting_amd/libmvfit.so /tmp/keep.so
PYTHONPATH=. timeout 600 python tests/phase_timing.py > gpurun_out/r3q/phases.log 2>&1
cp /tmp/keep.so mvsmplfitting_amd/libmvfit.so
grep -A8 "^vposer_sparse rounds" gpurun_out/r3q/phases.log
