# gpurun driver: A/B bit comparison (tools/ab_bits.py, three modes) + phase timing + bench lines of the current build
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; rm -rf $O; mkdir -p $O
timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so vposer 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python tools/ab_bits.py mvsmplfitting_amd/libmvfit_old.so mvsmplfitting_amd/libmvfit.so gmm 2>&1 | grep -v amdgpu.ids | tail -6
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py > $O/phase_timing.log 2>&1; sed -n 2,9p $O/phase_timing.log | cut -c1-600
for a in "" "" "--prior vposer"; do timeout 300 python bench.py $a --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d.get('closure_rounds_per_fit'), d.get('final_loss_median'))"; done
