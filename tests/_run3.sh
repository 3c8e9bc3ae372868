python -m pytest tests/test_gpu_sharded_fit.py tests/test_gpu_async.py -q -x 2>&1 | tail -5
mkdir -p gpurun_out/r2d
python bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; tail -c 3000 gpurun_out/r2d/bench.json; tail -3 gpurun_out/r2d/bench.err
python bench.py --config demo --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2d/bench_demo.json 2> gpurun_out/r2d/bench_demo.err; cat gpurun_out/r2d/bench_demo.json | cut -c 1-1500; tail -3 gpurun_out/r2d/bench_demo.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --single-device --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/bench_2rank_single_device.log 2>&1; tail -c 1200 gpurun_out/r2d/bench_2rank_single_device.log
bash tools/collect_profiles.sh r2d 2>&1 | tail -30
