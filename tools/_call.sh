set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
for v in "" _A _B _C; do
  echo "=== variant '$v'" >> gpurun_out/r3o/quick.log
  MVFIT_LIBRARY=$GRAFT_REPO_ROOT/mvsmplfitting_amd/libmvfit$v.so timeout 120 python tests/quick_async.py 32 >> gpurun_out/r3o/quick.log 2>&1
  MVFIT_LIBRARY=$GRAFT_REPO_ROOT/mvsmplfitting_amd/libmvfit$v.so timeout 120 python tests/quick_async.py 32 >> gpurun_out/r3o/quick.log 2>&1
done
timeout 600 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_async.py tests/test_gpu_closure.py tests/test_gpu_dropin.py tests/test_gpu_demo.py -q > gpurun_out/r3o/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3o/tests.log
grep -v amdgpu gpurun_out/r3o/quick.log; tail -4 gpurun_out/r3o/tests.log
