cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --config configs2 --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'], d['us_per_round'])"; done
python bench.py --config configs2 --round-mode chained --steps 5 --warmup 1 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chained', d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'], d['us_per_round'])"
timeout 1200 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sdf_term.py tests/test_gpu_sdf_cull.py tests/test_gpu_sdf.py -q -x > gpurun_out/sdf_tests.log 2>&1; tail -4 gpurun_out/sdf_tests.log
