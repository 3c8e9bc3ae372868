"""Developer tool (needs the -DMVFIT_LB_CHECK build, mvsmplfitting_amd/libmvfit_check.so): runs the staged fits of the bench
workload (L2 prior, VPoser, GMM, reuse flag) with every fast optimiser transition cross-checked on the device against the
general state machine - word for word - and prints the counters: transitions checked / mismatching words."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MVFIT_LIBRARY'] = os.path.join(ROOT, 'mvsmplfitting_amd', 'libmvfit_check.so')
import numpy as np
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
B, V = 32, 8
model = syn.make_body_model(0, skin_topk=4); cams = syn.make_camera_ring(V)
eng = MvFit(model, vposer=syn.make_vposer_decoder(), gmm=syn.gmm_constants(syn.make_gmm(), np.float32))
lib = eng._lib
fr = syn.make_frames(B, seed0=1000); xgt = np.zeros((B, 118), np.float32)
for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items(): xgt[:, a:b] = fr[k]
eng.set_problems(cams, np.zeros((B, V, 17, 2), np.float32), np.ones((B, V, 17), np.float32))
_, joints = eng.vertices(xgt)
gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=1007); eng.set_problems(cams, gt, conf)
x0 = np.zeros((B, 118), np.float32); x0[:, 85] = 1
out = (C.c_uint32 * 4)()
bad = 0
for name, flags in (('l2', 0), ('l2 sparse', _lib.F_SPARSE_VERTS), ('vposer', _lib.F_VPOSER | _lib.F_SPARSE_VERTS), ('gmm', _lib.F_PRIOR_GMM | _lib.F_SPARSE_VERTS),
                    ('l2 reuse', _lib.F_SPARSE_VERTS | _lib.F_REUSE_OUTER_VALUE)):
    lib.mvfit_debug_lb_check(out, 1)
    xf, st = eng.fit(x0, stage_weights(1536.0, flags=flags))
    lib.mvfit_debug_lb_check(out, 1)
    print('%-10s closures %6d | fast transitions checked %6d  mismatching words %d  (first word %d)' % (name, int(st['n_closure'].sum()), out[0], out[1], out[2]))
    bad += out[1]
print('OK' if bad == 0 else 'MISMATCH')
