set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o
rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
for m in fused unfused; do
if [ $m = unfused ]; then export MVFIT_SDF_UNFUSED=1; else unset MVFIT_SDF_UNFUSED; fi
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$m -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-pmc --config configs2 > $R/$O/bench_$m.json 2> $R/$O/stats_$m.log )
find $O/stats_$m -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$m.csv \;
rm -rf $O/stats_$m
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r3o/kernel_stats_$m.csv')))
print('$m')
for r in rows[:9]:
    print('  %-44s calls %6s avg %7.1f us min %7.1f total %8.2f ms' % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
