import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
V = 8
model = syn.make_body_model(0)
cams = syn.make_camera_ring(V)
eng = MvFit(model, vposer=syn.make_vposer_decoder())
fr = syn.make_frames(B)
xgt = np.zeros((B,118), np.float32)
for k,(a,b) in dict(betas=(0,10), global_orient=(10,13), body_pose=(13,82), transl=(82,85), scale=(85,86)).items(): xgt[:,a:b] = fr[k]
gt0 = np.zeros((B,V,17,2),np.float32); cf0 = np.ones((B,V,17),np.float32)
eng.set_problems(cams, gt0, cf0)
_, joints = eng.vertices(xgt)
gt, conf = syn.make_observations(joints.cpu().numpy(), cams)
eng.set_problems(cams, gt, conf)
x0 = np.zeros((B,118), np.float32); x0[:,85]=1
for name, flags in (('full',0), ('sparse',_lib.F_SPARSE_VERTS), ('vp_full',_lib.F_VPOSER), ('vp_sparse',_lib.F_VPOSER|_lib.F_SPARSE_VERTS)):
    stages = stage_weights(1536.0, flags=flags)
    for rep in range(2):
        torch.cuda.synchronize(); t=time.time()
        xf, st = eng.fit(x0, stages)
        torch.cuda.synchronize(); dt=time.time()-t
    ncl = st['n_closure'].cpu().numpy(); fl = st['final_loss'].cpu().numpy()
    print('%-10s B=%d fit %.1f ms  closures total %d max %d  -> %.0f closures/s ; rounds/s %.0f ; final loss med %.1f max %.1f' % (name, B, dt*1e3, ncl.sum(), ncl.max(), ncl.sum()/dt, ncl.max()/dt, np.median(fl), fl.max()))
# closure timing
stages = stage_weights(1536.0)
x = torch.tensor(x0, device='cuda')
for name, flags in (('closure full',0), ('closure sparse', _lib.F_SPARSE_VERTS)):
    w = dict(stages[0], flags=flags)
    for _ in range(3): eng.closure(x, w)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(200): eng.closure(x, w)
    torch.cuda.synchronize(); dt=(time.time()-t)/200
    print(name, '%.1f us per batched closure call' % (dt*1e6))
eng.profile(True)
w = dict(stages[0], flags=0)
for _ in range(50): eng.closure(x, w)
print(eng.profile_read())
