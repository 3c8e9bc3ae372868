# gpurun driver: tests/phase_timing.py on the -DMVFIT_TIMING build -> gpurun_out/r5h/phase_timing.log
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py > $O/phase_timing.log 2>&1; head -12 $O/phase_timing.log | cut -c1-700
