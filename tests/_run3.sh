python -m pytest tests/test_umeyama.py tests/test_gpu_init_guess.py tests/test_project.py -q 2>&1 | tail -15
