set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_umeyama.py tests/test_gpu_init_guess.py tests/test_init_guess_ref.py tests/test_gpu_sequence.py tests/test_gpu_sdf_term.py tests/test_gpu_large_batch.py -q -m gpu > gpurun_out/r3b/new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/new_tests.log
tail -40 gpurun_out/r3b/new_tests.log
