import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
model = syn.make_body_model(0, skin_topk=4)
eng = MvFit(model, device=0)
cams, gt, conf, x0 = bench.build_inputs(eng, 32, 4, seed0=1000)
stages = stage_weights(1536.0, flags=0)
x0_d = torch.as_tensor(x0, device='cuda')
for _ in range(2): eng.fit(x0_d, stages)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): xf, st = eng.fit(x0_d, stages)
    torch.cuda.synchronize(); print('ms/fit', round((time.perf_counter() - t0) / 8 * 1e3, 3), st.get('passes'), int(st['n_closure'].sum()))
