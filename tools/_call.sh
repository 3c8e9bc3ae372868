set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3f/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/gpu_tests.log
grep -E "passed|failed|rc=" gpurun_out/r3f/gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3f/smoke.log 2>&1; tail -1 gpurun_out/r3f/smoke.log
O=gpurun_out/r3g
rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o s -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants --no-pmc > $R/$O/bench_under_rocprof.json 2> $R/$O/stats.log )
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_sdf -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-pmc --config configs2 > $R/$O/bench_sdf_under_rocprof.json 2> $R/$O/stats_sdf.log )
find $O/stats_sdf -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_sdf.csv \;
rm -rf $O/stats $O/stats_sdf
timeout 900 python bench.py > $O/bench.json.log 2> $O/bench.err
B="timeout 400 python bench.py --no-cpu-baseline --no-pmc --no-variants"
$B --config configs2 > $O/bench_sdf.json.log 2>&1
$B --config configs3 > $O/bench_configs3.json.log 2>&1
$B --config configs4 > $O/bench_configs4.json.log 2>&1
$B --config demo > $O/bench_demo.json.log 2>&1
$B --prior vposer > $O/bench_vposer.json.log 2>&1
$B --sparse > $O/bench_sparse.json.log 2>&1
MVFIT_ROUND_MODE=serial $B > $O/bench_chained.json.log 2>&1
python - <<'PY'
import json, glob
for fn in sorted(glob.glob('gpurun_out/r3g/bench*.log')):
    try:
        l=[x for x in open(fn) if x.startswith('{')]
        d=json.loads(l[-1]); r=d.get('roofline') or {}
        print(fn.split('/')[-1], d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d.get('vertex_passes_lost_in_timed_fits'), d.get('decoder_helpers_last_fit'), r.get('avg_launch_us'), r.get('frac'), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(fn, 'failed', e); print(open(fn).read()[-800:])
PY
