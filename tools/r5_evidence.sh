# gpurun driver: profiles/r5_ab_bits.log, r5_phase_timing.log, r5_kernel_stats_configs3.csv
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
( echo "# tools/ab_bits.py: libmvfit_old.so = the build of commit 0f1ae9a (resident pass, before the optimiser-kernel steps) vs the final build"
  for m in "" vposer gmm; do echo "## mode: ${m:-l2}"; timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so $m 2>&1 | grep -v amdgpu.ids; done ) > $O/ab_bits.log 2>&1
tail -3 $O/ab_bits.log
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_timing.log; sed -n 1,12p $O/phase_timing.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats3 -o s -- python $GRAFT_REPO_ROOT/bench.py --config configs3 --steps 5 --no-variants --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_configs3_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/stats3 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_configs3.csv; head -4 $O/kernel_stats_configs3.csv | cut -c1-220
