#!/bin/bash
# Run on the MI355X box (gpurun): the bench lines of a round.  Usage: bash tools/collect_round.sh <tag> [quick]
#   -> gpurun_out/<tag>/bench_*.json.log (copy what is to be judged into profiles/<tag>_*)
set -u
TAG=${1:-r6}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
NOX="--no-pmc --no-cpu-baseline --no-variants"
timeout 900 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 $NOX > $OUT/bench_driver_command.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --sparse $NOX > $OUT/bench_sparse.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --prior vposer $NOX > $OUT/bench_vposer.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --prior vposer --sparse $NOX > $OUT/bench_vposer_sparse.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --prior gmm $NOX > $OUT/bench_gmm.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --config configs3 --prior vposer $NOX > $OUT/bench_configs3_vposer.json.log 2>> $OUT/bench.err
timeout 300 python bench.py $NOX --round-mode chained > $OUT/bench_chained.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --gpus 2 --dist-backend gloo --single-device $NOX > $OUT/bench_2rank_gloo_single_device.json.log 2>> $OUT/bench.err
timeout 600 python bench.py --config configs2 --sdf-faces all --steps 3 $NOX > $OUT/bench_sdf_all_faces.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --config demo $NOX > $OUT/bench_demo.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --config configs3 --no-cpu-baseline --no-variants > $OUT/bench_configs3.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --config configs4 --no-cpu-baseline --no-variants > $OUT/bench_configs4.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --resident-pass 0 $NOX > $OUT/bench_per_round_launches.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --frames 128 $NOX > $OUT/bench_b128.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --frames 256 $NOX > $OUT/bench_b256.json.log 2>> $OUT/bench.err
timeout 600 python bench.py --config configs2 --no-pmc --no-cpu-baseline > $OUT/bench_sdf.json.log 2>> $OUT/bench.err
timeout 300 python bench.py --config folder --steps 3 > $OUT/bench_folder.json.log 2>> $OUT/bench.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench*.json.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(os.path.basename(f), d.get("value"), d.get("unit"), d.get("ms_per_step"), d.get("closure_rounds_per_fit"), (d.get("roofline") or {}).get("frac"))
PY
