set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for CN in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $CN | tr ' ' '_')
  PYTHONPATH=$R timeout 300 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $R/$O/$tag -o p -- python $R/tools/pmc_vertex_pass.py drive 256 20 > /dev/null 2> $R/$O/$tag.log
done
cd $R
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda:[0,0.0])
for fn in glob.glob('gpurun_out/r3t/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(fn)):
        if 'vertex_pass' not in row['Kernel_Name']: continue
        k=(row['Kernel_Name'].split('(')[0][-30:], row['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(row['Counter_Value'])
for k,(n,t) in sorted(acc.items()): print(k, n, t/n)
PY
tail -3 $O/*.log | grep -i "error\|invalid\|not" | head
