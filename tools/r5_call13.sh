set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; rm -rf $O; mkdir -p $O
for m in "" vposer gmm; do timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so $m 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 900 python -m pytest tests -m gpu -q -x -k "not real_caller and not sdf_term_last_stage and not all_faces" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|Error|assert" $O/tests.log | tail -5
