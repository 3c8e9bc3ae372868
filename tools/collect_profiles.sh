#!/bin/bash
# Run on the MI355X box (gpurun): kernel stats of the bench command + PMC traffic of the vertex pass at 32 / 128
# problems.  Usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...  (copy the summaries into profiles/)
set -u
TAG=${1:-r3}
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
cd $R
python tools/pmc_vertex_pass.py parse $OUT/pmc.json B32=$OUT/pmc_b32_FETCH_SIZE B32=$OUT/pmc_b32_WRITE_SIZE B32=$OUT/pmc_b32_SQ B128=$OUT/pmc_b128_FETCH_SIZE B128=$OUT/pmc_b128_WRITE_SIZE B128=$OUT/pmc_b128_SQ
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
