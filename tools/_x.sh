cd $GRAFT_REPO_ROOT
O=gpurun_out/sdfsvc; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --config configs2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-variants > $O/bench_configs2.json.log 2> $O/bench_configs2.err; tail -3 $O/bench_configs2.err; python -c "
import json; d=json.loads(open('$O/bench_configs2.json.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'], d['vertex_passes_last_fit'])"
timeout 1200 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sdf_term.py tests/test_gpu_sdf_cull.py -q -s -k "vposer_fits_with_the_sdf" > $O/tests.log 2>&1; tail -15 $O/tests.log
