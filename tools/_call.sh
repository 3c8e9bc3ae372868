set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p2
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sdf_term.py tests/test_gpu_sdf.py -q > $O/tests.log 2>&1; grep -E "passed|failed|Error" $O/tests.log | tail -3
B="timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-variants"
for rep in 1 2; do $B --config configs2 > $O/sdf_$rep.log 2>&1; done
python - <<'PY'
import json, glob
for fn in sorted(glob.glob('gpurun_out/r3p2/sdf*.log')):
    try:
        l=[x for x in open(fn) if x.startswith('{')]
        d=json.loads(l[-1]); print(fn.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'])
    except Exception as e:
        print(fn, 'failed', e); print(open(fn).read()[-800:])
PY
