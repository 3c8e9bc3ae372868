# Developer tool: A/B of library variants on ONE GPU box (box-to-box spread is ~5 %: only runs of one call compare).
# Build the variants next to the product, e.g.
#   (cd mvsmplfitting_amd/csrc && make -j8 OBJDIR=build_x OUT=../libmvfit_x.so EXTRA=-DSOMETHING)
#   git stash; (cd mvsmplfitting_amd/csrc && make -j8 OBJDIR=build_base OUT=../libmvfit_base.so); git stash pop
# then  gpurun -- 'VARIANTS="base x" bash tools/ab_variants.sh'
# Prints value, ms per fit, closure rounds, final-loss median (rounds + median = the bit-identity indicator) and us per round.
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py "$@" --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LBL $*', d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'], d['us_per_round'])"; }
for rep in 1 2; do
for v in ${VARIANTS:-base}; do
  LBL=$v; export MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_$v.so
  run --frames 32; run --prior vposer
done; done
