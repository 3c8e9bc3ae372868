set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_sdf_term.py tests/test_gpu_sdf.py tests/test_gpu_dropin.py -q > gpurun_out/r3n/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3n/tests.log
timeout 300 python bench.py --config configs2 --no-cpu-baseline --no-pmc > gpurun_out/r3n/bench_configs2.log 2>&1
MVFIT_SDF_ONE_PHASE=1 timeout 300 python bench.py --config configs2 --no-cpu-baseline --no-pmc > gpurun_out/r3n/bench_configs2_onephase.log 2>&1
tail -6 gpurun_out/r3n/tests.log
for f in gpurun_out/r3n/bench*.log; do python - "$f" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[1],'NO JSON', open(sys.argv[1]).read()[-800:])
else:
    d=json.loads(l[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], 'rounds', d['closure_rounds_per_fit'], 'cl/frame', d['closures_per_fit_per_frame'], 'final', d['final_loss_median'], d['vertex_passes_last_fit'])
PY
done
