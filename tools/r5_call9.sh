set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "not real_caller and not sdf_term_last_stage and not all_faces" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=|Error|assert" $O/tests.log | tail -8
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so timeout 300 python tests/phase_timing.py > $O/phase_timing.log 2>&1; sed -n 2,9p $O/phase_timing.log | cut -c1-600
for a in "" "--prior gmm" "--prior vposer" "--config configs3" "--config demo"; do timeout 300 python bench.py $a --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d.get('closure_rounds_per_fit'), d.get('final_loss_median'))"; done
