cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gst -o s -- python $GRAFT_REPO_ROOT/bench.py --config configs2 --steps 3 --no-variants --no-pmc --no-cpu-baseline > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/gst -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-60,200-330
