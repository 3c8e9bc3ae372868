# gpurun driver: round-6 baseline of the build the round started from + issue-side counters of the role-split kernel
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6base; rm -rf $O; mkdir -p $O
rocm-smi --showclocks > $O/clocks.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json.log 2> $O/bench.err
for CFG in "--config configs3" "--prior vposer" "--prior vposer --vposer-sets 8" "--prior gmm" "--config configs2" "--frames 256" "--config demo" "--config configs4"; do
  N=$(echo $CFG | tr -d ' -'); timeout 400 python bench.py $CFG --steps 5 --warmup 1 --no-pmc --no-cpu-baseline --no-variants > $O/bench_$N.json.log 2> $O/bench_$N.err
done
bash tools/pmc_issue_resident.sh 128 r6issue > $O/issue.log 2>&1
bash tools/pmc_issue_resident.sh 32 r6issue32 > $O/issue32.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6base/bench*.json.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
