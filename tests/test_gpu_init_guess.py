"""GPU: the batched per-frame initial guess (mvsmplfitting_amd.init_guess.init_guess_batch = mvfit_triangulate +
mvfit_umeyama, reference code/utils/init_guess.py:18-106) on the reference's demo frame: the golden file holds what the
reference's own recompute3D + umeyama produced there (oracle/make_golden_demo.py; cv2.Rodrigues replaced by scipy's
conversion, cv2 being absent in the build container)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import init_guess as ig
from oracle import umeyama_np as un
from tests.gpu_helpers import make_engine
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu


def test_demo_initial_guess_equals_the_reference():
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    eng = make_engine(body_model())
    cams = tuple(g[k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    kps = g['keypoints'][None].astype(np.float32)                          # [1, 6, 17, 3]
    eng.set_problems(cams, kps[..., :2], kps[..., 2])
    out = ig.init_guess_batch(eng, g['extris'], g['intris'], kps, est_scale=True, use_torso=True)
    j3 = out['joints3d'][0].cpu().numpy()
    assert np.abs(j3 - g['init_joints3d']).max() < 1e-6 * np.abs(g['init_joints3d']).max()
    rest = ig.rest_keypoints(eng).cpu().numpy()
    assert np.abs(rest - g['init_joints_rest']).max() < 1e-5               # float32 forward vs the reference's float64
    # the guess, against the restatement evaluated on the device's own inputs (numpy's singular-vector signs: the
    # device SVD walks LAPACK's path, csrc/lapack_svd3.h)
    rot, trans, scale = out['rot'][0].cpu().numpy(), out['transl'][0].cpu().numpy(), float(out['scale'][0])
    t = list(ig.TORSO)
    r_ref, t_ref, s_ref, _ = un.umeyama(rest[t], j3[t], True)
    assert np.abs(rot - r_ref).max() < 1e-7 and np.abs(trans - t_ref).max() < 1e-7 and abs(scale - s_ref) < 1e-9
    # ... and against what the REFERENCE computed on this frame (x0 of the golden fit: the reference's own recompute3D +
    # umeyama + rotation vector): global_orient, transl, scale (float32 rest keypoints here, float64 there)
    x0 = g['x0']
    assert np.abs(out['global_orient'][0].cpu().numpy() - x0[10:13]).max() < 1e-4, (out['global_orient'][0], x0[10:13])
    assert np.abs(trans - x0[13:16]).max() < 1e-4 * max(1.0, np.abs(x0[13:16]).max())
    assert abs(scale - x0[16]) < 1e-4 * x0[16]
    # fix_params: the flat start of the fit
    x = ig.initial_params(out, use_vposer=True).cpu().numpy()
    assert x.shape == (1, 118) and np.all(x[0, 13:82] == 0) and abs(x[0, 85] - scale) < 1e-5
    x2 = ig.initial_params(out, use_vposer=False).cpu().numpy()
    assert np.all(x2[0, 13:19] == 1.0) and np.all(x2[0, 19:82] == 0)
    eng.close()


def test_batched_initial_guess_is_per_frame():
    """B frames of one rig in one call == the frames one by one."""
    d = dict(np.load(os.path.join(GOLD, 'triangulate.npz')))
    eng = make_engine(body_model())
    ext, intr, kps = d['v8_extris'], d['v8_intris'], d['v8_kps']
    B, V = kps.shape[0], kps.shape[1]
    cams = (ext[:, :3, :3].astype(np.float32), ext[:, :3, 3].astype(np.float32), intr[:, 0, 0].astype(np.float32),
            intr[:, :2, 2].astype(np.float32))
    eng.set_problems(cams, kps[..., :2], kps[..., 2])
    all_ = ig.init_guess_batch(eng, ext, intr, kps)
    for b in range(B):
        eng.set_problems(cams, kps[b:b + 1, ..., :2], kps[b:b + 1, ..., 2])
        one = ig.init_guess_batch(eng, ext, intr, kps[b:b + 1])
        for k in ('global_orient', 'transl', 'scale'):
            assert np.array_equal(one[k][0].cpu().numpy(), all_[k][b].cpu().numpy()), (b, k)
    eng.close()


def test_single_view_guess_and_fit():
    """One camera only (init_guess.py:54-72): the depth guess equals its restatement, and the staged single-view fit
    started from it ends far below its start.  (The branch is pinned to the reference's own init_guess in
    tests/test_init_guess_ref.py.)"""
    from mvsmplfitting_amd.engine import stage_weights
    from oracle import init_guess_np as ign
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    eng = make_engine(body_model())
    kp6 = g['keypoints'].reshape(6, 17, 3).astype(np.float32)
    kps = np.stack([kp6[0], kp6[3]])[:, None]                               # two "frames" seen by camera 0 only: [2, 1, 17, 3]
    cams = tuple(g[k][:1].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    eng.set_problems(cams, kps[..., :2], kps[..., 2])
    out = ig.init_guess_batch(eng, g['extris'][:1], g['intris'][:1], kps, est_scale=True, use_torso=True)
    rest = ig.rest_keypoints(eng).cpu().numpy()
    for b in range(2):
        ref = ign.single_view_joints3d(rest, g['extris'][0], g['intris'][0], kps[b, 0])
        assert np.abs(out['joints3d'][b].cpu().numpy() - ref).max() < 1e-9 * np.abs(ref).max()
    x0 = ig.initial_params(out, use_vposer=False)
    stages = stage_weights(1536.0, flags=0)
    l0 = eng.closure(x0, dict(stages[-1]), want_grad=False)['loss'].cpu().numpy()
    xf, st = eng.fit(x0, stages)
    final = st['final_loss'].cpu().numpy()
    assert np.all(np.isfinite(final)) and np.all(final < 0.5 * l0), (final, l0)
    eng.close()
