set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 400 python tests/vp_timeline.py 256 > gpurun_out/r3h/vp_timeline.log 2>&1
tail -8 gpurun_out/r3h/vp_timeline.log
