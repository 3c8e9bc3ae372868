# round 6: the bench lines + profiles of the final build (gpurun_out/r6/...)
set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_round.sh r6 > gpurun_out/r6_collect_round.log 2>&1
tail -25 gpurun_out/r6_collect_round.log
NOX="--no-pmc --no-cpu-baseline --no-variants"
timeout 300 python bench.py --frames 256 --work-queue 0 $NOX > gpurun_out/r6/bench_b256_sub_batches.json.log 2>> gpurun_out/r6/bench.err
timeout 300 python bench.py --frames 512 $NOX > gpurun_out/r6/bench_b512.json.log 2>> gpurun_out/r6/bench.err
timeout 300 python bench.py --prior vposer --resident-pass 0 --vposer-sets 16 $NOX > gpurun_out/r6/bench_vposer_16_sets_per_round_launches.json.log 2>> gpurun_out/r6/bench.err
timeout 600 python bench.py --config configs2 $NOX > gpurun_out/r6/bench_sdf_service.json.log 2>> gpurun_out/r6/bench.err
bash tools/collect_profiles.sh r6p > gpurun_out/r6_collect_profiles.log 2>&1
tail -30 gpurun_out/r6_collect_profiles.log
bash tools/pmc_issue_resident.sh 128 r6issue > gpurun_out/r6/issue128.log 2>&1
bash tools/pmc_issue_resident.sh 32 r6issue32 > gpurun_out/r6/issue32.log 2>&1
for B in 32 128; do MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 200 python tests/vp_resident_timeline.py $B > gpurun_out/r6/resident_timeline_$B.log 2>&1; done
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py > gpurun_out/r6/phase_timing.log 2>&1
timeout 300 python tools/predict_scaling.py configs1 gpurun_out/r6/predicted_scaling_configs1.json > gpurun_out/r6/predict1.log 2>&1
timeout 300 python tools/predict_scaling.py configs3 gpurun_out/r6/predicted_scaling_configs3.json > gpurun_out/r6/predict3.log 2>&1
cd /tmp && export TMPDIR=/tmp
for C in configs2 configs3; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6/stats_$C -o s -- python $GRAFT_REPO_ROOT/bench.py --config $C --steps 5 --no-variants --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r6/bench_${C}_under_rocprof.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r6/stats_$C -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r6/kernel_stats_$C.csv
done
cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/r6/stats_configs2 gpurun_out/r6/stats_configs3 gpurun_out/r6p/stats
for d in gpurun_out/r6p gpurun_out/r6issue gpurun_out/r6issue32; do find $d -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} + ; done
ls gpurun_out/r6 | head -60
