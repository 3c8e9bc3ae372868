set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py -q -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 200 python tests/vp_resident_timeline.py 128 > $O/timeline_128.log 2>&1; tail -6 $O/timeline_128.log
timeout 300 python bench.py --config configs3 --no-cpu-baseline --no-pmc > $O/bench_configs3.log 2>&1; tail -1 $O/bench_configs3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("slowest_workgroup_us"), r.get("alone_per_round_us"))"
timeout 300 python bench.py --config configs4 --no-cpu-baseline --no-pmc --no-variants > $O/bench_configs4.log 2>&1; tail -1 $O/bench_configs4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("slowest_workgroup_us"), r.get("alone_per_round_us"))"
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-variants > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r.get('slowest_workgroup_us'), r.get('alone_per_round_us'))"
