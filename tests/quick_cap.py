"""Developer tool: time a fit with a fixed number of closure rounds (max_rounds cap) - equal work for A/B / ablation builds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights, MvFitError
B, V, CAP = int(os.environ.get('QC_B', 32)), 8, int(sys.argv[1]) if len(sys.argv) > 1 else 240
MODE = os.environ.get('QC_MODE', 'sparse')      # sparse | full | vposer (sparse + VPoser prior)
model = syn.make_body_model(0, skin_topk=4); cams = syn.make_camera_ring(V)
eng = MvFit(model, vposer=syn.make_vposer_decoder() if MODE == 'vposer' else None)
fr = syn.make_frames(B, seed0=1000); xgt = np.zeros((B, 118), np.float32)
for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items(): xgt[:, a:b] = fr[k]
eng.set_problems(cams, np.zeros((B, V, 17, 2), np.float32), np.ones((B, V, 17), np.float32))
_, joints = eng.vertices(xgt)
gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=1007); eng.set_problems(cams, gt, conf)
x0 = np.zeros((B, 118), np.float32); x0[:, 85] = 1
stages = stage_weights(1536.0, flags={'sparse': _lib.F_SPARSE_VERTS, 'full': 0, 'vposer': _lib.F_SPARSE_VERTS | _lib.F_VPOSER}[MODE])
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t = time.time()
    try:
        eng.fit(x0, stages, max_rounds=CAP)
    except MvFitError:
        pass
    torch.cuda.synchronize(); best = min(best, time.time() - t)
print(MODE, B, '%s: %d rounds %.3f ms -> %.2f us per round' % (os.path.basename(os.environ.get('MVFIT_LIBRARY', 'libmvfit.so')), CAP, best * 1e3, best * 1e6 / CAP))
