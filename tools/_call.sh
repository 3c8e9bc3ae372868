set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i
mkdir -p $O
B="timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-variants"
timeout 600 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_trajectory.py tests/test_gpu_reuse.py -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
$B > $O/default_$rep.log 2>&1
$B --sparse > $O/sparse_$rep.log 2>&1
done
$B --frames 128 > $O/b128.log 2>&1
$B --prior vposer > $O/vposer.log 2>&1
python - <<'PY'
import json, glob
for fn in sorted(glob.glob('gpurun_out/r3i/*.log')):
    try:
        l=[x for x in open(fn) if x.startswith('{')]
        d=json.loads(l[-1]); r=d.get('roofline') or {}
        print(fn.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], r.get('avg_launch_us'))
    except Exception as e:
        pass
PY
cp mvsmplfitting_amd/libmvfit.so /tmp/keep.so
PYTHONPATH=. timeout 600 python tests/phase_timing.py > $O/phases.log 2>&1
cp /tmp/keep.so mvsmplfitting_amd/libmvfit.so
grep -A7 "^sparse rounds" $O/phases.log
