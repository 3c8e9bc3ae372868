mkdir -p gpurun_out/r2e
python bench.py --frames 128 --steps 4 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2e/bench_b128.json 2> gpurun_out/r2e/err1
MVFIT_ROUND_MODE=serial python bench.py --frames 128 --steps 4 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2e/bench_b128_serial.json 2> gpurun_out/r2e/err2
python bench.py --frames 128 --steps 4 --warmup 1 --no-cpu-baseline --no-variants --sparse > gpurun_out/r2e/bench_b128_sparse.json 2> gpurun_out/r2e/err3
python bench.py --frames 64 --steps 4 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2e/bench_b64.json 2> gpurun_out/r2e/err4
python bench.py --views 16 --steps 5 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2e/bench_v16.json 2> gpurun_out/r2e/err5
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/err6
python - <<'PY'
import json
for n in ('b128','b128_serial','b128_sparse','b64','v16','default'):
    try:
        d=json.load(open('gpurun_out/r2e/bench_%s.json'%n)); r=d.get('roofline') or {}
        print(n, d['value'], d['ms_per_step'], d['closure_rounds_per_fit'], d['closures_per_fit_per_frame'], d.get('vertex_passes_last_fit'), r.get('avg_launch_us'), r.get('alone_back_to_back_us'), r.get('frac'))
    except Exception as e: print(n,'ERR',e)
PY
tail -2 gpurun_out/r2e/err1
