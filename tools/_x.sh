cd $GRAFT_REPO_ROOT
O=gpurun_out/sdfsvc; rm -rf $O; mkdir -p $O
for CFG in "--config configs2" "--frames 32" "--prior gmm"; do N=$(echo $CFG | tr -d ' -'); timeout 300 python bench.py $CFG --steps 5 --warmup 1 --no-pmc --no-cpu-baseline --no-variants > $O/bench_$N.json.log 2> $O/bench_$N.err; python -c "
import json; d=json.loads(open('$O/bench_$N.json.log').read().strip().splitlines()[-1]); print('$N', d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['final_loss_median'], d['us_per_round'], d['vertex_passes_last_fit'])"; done
timeout 1200 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_sdf_term.py -q -x -k "sdf or service" > $O/tests.log 2>&1; tail -5 $O/tests.log
