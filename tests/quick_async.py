"""Developer tool: asynchronous fit at several batch sizes - time, closures/s, vertex-pass counters."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
sizes = [int(a) for a in sys.argv[1:]] or [32, 128, 161, 256]
V = 8
model = syn.make_body_model(0, skin_topk=4)
cams = syn.make_camera_ring(V)
for B in sizes:
    eng = MvFit(model)
    fr = syn.make_frames(B, seed0=1000)
    xgt = np.zeros((B, 118), np.float32)
    for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items():
        xgt[:, a:b] = fr[k]
    eng.set_problems(cams, np.zeros((B, V, 17, 2), np.float32), np.ones((B, V, 17), np.float32))
    _, joints = eng.vertices(xgt)
    gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=1007)
    eng.set_problems(cams, gt, conf)
    x0 = np.zeros((B, 118), np.float32); x0[:, 85] = 1
    for name, flags in (('full', 0), ('sparse', _lib.F_SPARSE_VERTS)):
        stages = stage_weights(1536.0, flags=flags)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            xf, st = eng.fit(x0, stages)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        ncl = st['n_closure'].cpu().numpy()
        print('B=%4d %-7s fit %7.2f ms  closures %7d max %4d -> %9.0f closures/s  passes %s  alone-pass %.1f us' % (
            B, name, best * 1e3, ncl.sum(), ncl.max(), ncl.sum() / best, st['passes'],
            1e3 * eng.profile_vertex_pass_ms(32) if name == 'full' else 0.0), flush=True)
    eng.close()
