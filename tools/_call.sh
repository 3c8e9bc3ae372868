set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 2 > $O/bench_driver_cmd.log 2> $O/bench_driver_cmd.err; T1=$(date +%s.%N); echo "wall $(echo "$T1 - $T0" | bc) s"; grep -c "^{" $O/bench_driver_cmd.log; wc -l $O/bench_driver_cmd.log; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3u/bench_driver_cmd.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['roofline']['traffic_source'][:60], d['cpu_baseline']['value'])
PY
