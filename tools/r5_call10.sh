set -u
cd $GRAFT_REPO_ROOT
timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so 2>&1 | grep -v amdgpu.ids | tail -20
