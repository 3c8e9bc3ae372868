# gpurun driver: LDS counters of the stand-alone resident dispatches at 128 problems (role-split kernel) -> gpurun_out/r5lds/pmc_lds_resident.json
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=128
for CN in SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS; do
  timeout 200 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $OUT/pmc_$CN -o p -- python $R/tools/pmc_vertex_pass.py drive_resident $B $OUT/pmc_$CN.rounds.json > /dev/null 2> $OUT/pmc_$CN.log || echo "$CN failed"
done
cd $R
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from tools import pmc_vertex_pass as pv
e = {}
for cn in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS"):
    try:
        side = json.load(open("$OUT/pmc_%s.rounds.json" % cn))
        v, used = pv.resident_per_round("$OUT/pmc_%s" % cn, cn, side["rounds_per_standalone_dispatch"])
        e[cn + "_per_round"] = v
        e["workgroups_x_tiles"] = side.get("tiles_per_workgroup")
    except Exception as ex:
        e[cn] = "unavailable: %r" % (ex,)
json.dump(e, open("$OUT/pmc_lds_resident.json", "w"), indent=1, sort_keys=True)
print(json.dumps(e, indent=1, sort_keys=True))
PY
