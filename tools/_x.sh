cd $GRAFT_REPO_ROOT
for L in libmvfit.so libmvfit_s32.so libmvfit.so libmvfit_s32.so; do for CFG in "--prior vposer" "--frames 32" "--config configs3"; do MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/$L python bench.py $CFG --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$L', '$CFG', d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'))"; done; done
