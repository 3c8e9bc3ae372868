"""Developer tool: work counters of the all-faces SDF term on face lists (needs libmvfit_sdfstats.so:
make OUT=../libmvfit_sdfstats.so OBJDIR=build_sdfstats EXTRA=-DMVFIT_SDF_STATS; MVFIT_LIBRARY points at it) and the time of one closure."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
B, V, G = 32, 8, 128
model = syn.make_body_model(0, skin_topk=4); cams = syn.make_camera_ring(V)
eng = MvFit(model)
fr = syn.make_frames(B, seed0=1000); x = np.zeros((B, 118), np.float32)
for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items(): x[:, a:b] = fr[k]
eng.set_problems(cams, np.zeros((B, V, 17, 2), np.float32), np.ones((B, V, 17), np.float32))
eng.set_sdf(model['faces'], num_faces=None, grid_size=G)
w = dict(stage_weights(1536.0, coll_w=[0.0, 0.0, 1000.0, 4500.0])[3])
lib = eng._lib
st = (C.c_ulonglong * 8)()
has = hasattr(lib, 'mvfit_debug_sdf_stats')
for rep in range(3):
    if has: lib.mvfit_debug_sdf_stats(st, 1)
    torch.cuda.synchronize(); t = time.time()
    out = eng.closure(x, w, want_grad=True)
    torch.cuda.synchronize(); dt = time.time() - t
smp, S = eng.sdf_term_read()
smp = smp.cpu().numpy()
print('closure with the term: %.3f ms; S = %s' % (dt * 1e3, S.cpu().numpy()[:4]))
print('loss', out['loss'].cpu().numpy()[:4], 'nan in grad', np.isnan(out['grad'].cpu().numpy()).sum(), 'nan in samples', np.isnan(smp).sum(),
      'inside vertices per problem', (smp[..., 0] != 0).sum(1)[:6])
if has:
    lib.mvfit_debug_sdf_stats(st, 0)
    n = [int(v) for v in st]
    print('per problem: corners in range %.0f, ray tests %.0f (%.1f per corner, %.1f skipped by depth), inside corners %.0f, distance tests %.0f (%.1f per inside corner), corners that walked all faces %.0f'
          % (n[0] / B, n[1] / B, n[1] / max(1, n[0]), n[5] / max(1, n[0]), n[2] / B, n[3] / B, n[3] / max(1, n[2]), n[4] / B))
