set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; rm -rf $O; mkdir -p $O
cp mvsmplfitting_amd/libmvfit.so /tmp/keep.so
for B in 128 256; do PYTHONPATH=. timeout 300 python tests/vp_timeline.py $B > $O/tl_$B.log 2>&1; grep -v amdgpu $O/tl_$B.log | tail -4; done
cp /tmp/keep.so mvsmplfitting_amd/libmvfit.so
