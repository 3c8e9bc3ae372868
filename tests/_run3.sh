python -m pytest tests/test_gpu_closure.py -q -x -k "half_width or split_fp16" 2>&1 | tail -5
MVFIT_HALF_BASIS=1 PYTHONPATH=. python tests/report_vertex_pass.py 32 128
