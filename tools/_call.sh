set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_trajectory.py tests/test_gpu_lbfgs.py tests/test_gpu_large_batch.py tests/test_gpu_closure.py -q > $O/tests.log 2>&1; tail -2 $O/tests.log
B="timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-variants"
for rep in 1 2 3; do
  $B > $O/default_$rep.log 2>&1
  $B --sparse > $O/sparse_$rep.log 2>&1
done
$B --prior gmm > $O/gmm.log 2>&1
$B --config configs3 > $O/configs3.log 2>&1
MVFIT_ROUND_MODE=serial $B > $O/chained.log 2>&1
$B --prior vposer > $O/vposer.log 2>&1
$B --config configs2 > $O/sdf.log 2>&1
python - <<'PY'
import json, glob
for fn in sorted(glob.glob('gpurun_out/r3j/*.log')):
    try:
        l=[x for x in open(fn) if x.startswith('{')]
        d=json.loads(l[-1]); r=d.get('roofline') or {}
        print(fn.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], r.get('avg_launch_us'))
    except Exception as e:
        pass
PY
