# gpurun driver: ISSUE-side counters of the stand-alone resident dispatches (default 128 problems: the role-split kernel)
#   -> gpurun_out/$TAG/pmc_issue_resident.json   (per closure round, summed over the chip; bash tools/pmc_issue_resident.sh [B] [TAG])
set -u
R=$GRAFT_REPO_ROOT; B=${1:-128}; TAG=${2:-r6issue}; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR"
         "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
         "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"
         "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES")
run() {   # $1 = directory tag, rest = counters of one pass
  local t=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$t -o p -- python $R/tools/pmc_vertex_pass.py drive_resident $B $OUT/pmc_$t.rounds.json > /dev/null 2> $OUT/pmc_$t.log
}
i=0
for G in "${GROUPS_[@]}"; do
  if run g$i $G; then echo "g$i: $G" >> $OUT/groups.txt
  else
    echo "group g$i failed as one pass: one counter per pass" >> $OUT/groups.txt
    for CN in $G; do run $CN $CN && echo "$CN: $CN" >> $OUT/groups.txt || echo "$CN failed" >> $OUT/groups.txt; done
  fi
  i=$((i+1))
done
cd $R
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from tools import pmc_vertex_pass as pv
e = {"problems": $B}
for line in open("$OUT/groups.txt"):
    if ":" not in line or "failed" in line:
        continue
    tag, cns = line.strip().split(":", 1)
    for cn in cns.split():
        try:
            side = json.load(open("$OUT/pmc_%s.rounds.json" % tag))
            v, used = pv.resident_per_round("$OUT/pmc_%s" % tag, cn, side["rounds_per_standalone_dispatch"])
            e[cn + "_per_round"] = v
            e["tiles_per_workgroup_form"] = side.get("tiles_per_workgroup")
        except Exception as ex:
            e[cn] = "unavailable: %r" % (ex,)
json.dump(e, open("$OUT/pmc_issue_resident.json", "w"), indent=1, sort_keys=True)
print(json.dumps(e, indent=1, sort_keys=True))
PY
