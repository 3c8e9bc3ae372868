set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k
rm -rf $O; mkdir -p $O
B="timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-variants"
for rep in 1 2; do
for c in k28 k8 k12 k16; do
  L=$PWD/mvsmplfitting_amd/libmvfit_$c.so
  MVFIT_LIBRARY=$L $B > $O/default_${c}_$rep.log 2>&1
  MVFIT_LIBRARY=$L $B --sparse > $O/sparse_${c}_$rep.log 2>&1
done
done
python - <<'PY'
import json, glob
for fn in sorted(glob.glob('gpurun_out/r3k/*.log')):
    try:
        l=[x for x in open(fn) if x.startswith('{')]
        d=json.loads(l[-1]); r=d.get('roofline') or {}
        print(fn.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], r.get('avg_launch_us'))
    except Exception as e:
        print(fn, 'failed', e); print(open(fn).read()[-800:])
PY
