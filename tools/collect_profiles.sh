#!/bin/bash
# Run on the MI355X box (gpurun): kernel stats of the bench command + PMC traffic of the vertex pass at 32 / 128
# problems.  Usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...  (copy the summaries into profiles/)
set -u
TAG=${1:-r6}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants --no-pmc > $OUT/bench_under_rocprof.json 2> $OUT/stats.log   # (event stamps of the in-fit launches are distorted under the profiler: use the csv for durations, a plain run for the JSON line)
for B in 32 128; do
  for CN in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $OUT/pmc_b${B}_$CN -o p -- python $R/tools/pmc_vertex_pass.py drive $B 40 > /dev/null 2> $OUT/pmc_b${B}_$CN.log
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_b${B}_SQ -o p -- python $R/tools/pmc_vertex_pass.py drive $B 40 > /dev/null 2> $OUT/pmc_b${B}_SQ.log
done
# the RESIDENT pass (round 5): one fit's resident dispatch + three stand-alone ones serving 100 rounds each; every dispatch of the
# kernel summed, rounds served in the side file (bench.py: measure_pmc_resident does the same inside the default run)
for B in 32 128; do
  for CN in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $OUT/pmc_res_b${B}_$CN -o p -- python $R/tools/pmc_vertex_pass.py drive_resident $B $OUT/pmc_res_b${B}_$CN.rounds.json > /dev/null 2> $OUT/pmc_res_b${B}_$CN.log
  done
done
cd $R
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from tools import pmc_vertex_pass as pv
res = {}
for B in (32, 128):
    e = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
        side = json.load(open("$OUT/pmc_res_b%d_%s.rounds.json" % (B, cn)))
        v, used = pv.resident_per_round("$OUT/pmc_res_b%d_%s" % (B, cn), cn, side["rounds_per_standalone_dispatch"])
        e[cn + ("_KiB_per_round" if cn != "SQ_VALU_MFMA_BUSY_CYCLES" else "_per_round")] = v
        e["standalone_dispatches"] = used; e["rounds_per_dispatch"] = side["rounds_per_standalone_dispatch"]
        e["tiles_per_workgroup"] = side["tiles_per_workgroup"]
    e["traffic_bytes_per_round"] = (2.0 * e["FETCH_SIZE_KiB_per_round"] + e["WRITE_SIZE_KiB_per_round"]) * 1024.0
    e["note"] = "stand-alone dispatches of the resident kernel (lbs_vertex_pass_resident_kernel / _roles_kernel; 100 rounds each from the ring a fit left behind); 2 x FETCH_SIZE + WRITE_SIZE; the fit's own dispatch is left out (rocprofv3 serialises kernels while collecting counters)"
    res["RES_B%d" % B] = e
json.dump(res, open("$OUT/pmc_resident.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
python tools/pmc_vertex_pass.py parse $OUT/pmc.json B32=$OUT/pmc_b32_FETCH_SIZE B32=$OUT/pmc_b32_WRITE_SIZE B32=$OUT/pmc_b32_SQ B128=$OUT/pmc_b128_FETCH_SIZE B128=$OUT/pmc_b128_WRITE_SIZE B128=$OUT/pmc_b128_SQ
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
