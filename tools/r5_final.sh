# round 5: the bench lines + profiles of the final build (gpurun_out/r5/...)
set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_round.sh r5 > gpurun_out/r5_collect_round.log 2>&1
tail -25 gpurun_out/r5_collect_round.log
bash tools/collect_profiles.sh r5p > gpurun_out/r5_collect_profiles.log 2>&1
tail -30 gpurun_out/r5_collect_profiles.log
for B in 32 128; do MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 200 python tests/vp_resident_timeline.py $B > gpurun_out/r5/resident_timeline_$B.log 2>&1; done
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py > gpurun_out/r5/phase_timing.log 2>&1
