for nt in 0 1 2 3; do
MVFIT_DEBUG_NT_OFF=$nt python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants > /tmp/b.json 2>/dev/null
python - <<PY
import json
d=json.load(open('/tmp/b.json')); r=d['roofline']
print('nt_off_mask',$nt, d['value'], d['ms_per_step'], 'in-fit pass us', r['avg_launch_us'], 'alone', r['alone_back_to_back_us'])
PY
done
MVFIT_DEBUG_NT_OFF=1 python bench.py --steps 4 --warmup 1 --prior vposer --no-cpu-baseline --no-variants | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vposer ntoff1', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_sequence.py -q -x 2>&1 | tail -20
