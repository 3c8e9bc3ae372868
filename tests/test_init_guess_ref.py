"""The per-frame initial guess against the reference's OWN `init_guess` (code/utils/init_guess.py:18-114) run on the
shipped demo's cameras / keypoints (tests/golden/init_guess_ref.npz, oracle/make_golden_init_guess.py): three- and
six-view triangulation + umeyama, the single-view depth guess (:54-78) on two cameras, fixed and estimated scale.
CPU: the NumPy restatements (oracle/triangulate_np.py, init_guess_np.py, umeyama_np.py) reproduce it to 1e-9.
GPU: mvsmplfitting_amd.init_guess.init_guess_batch (mvfit_triangulate / the depth guess + mvfit_umeyama with LAPACK's
singular-vector signs) reproduces global_orient / transl / scale to 1e-4 (its rest keypoints come from the float32
device forward, the reference's from its float64 model)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import init_guess_np as ign
from oracle import triangulate_np as tn
from oracle import umeyama_np as un
from tests.helpers import GOLD, body_model

CASES = ['views6', 'views3', 'views6_fixscale', 'single0', 'single3', 'single0_fixscale']
TORSO = [5, 6, 11, 12]


def _load():
    g = np.load(os.path.join(GOLD, 'init_guess_ref.npz'))
    d = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    model = body_model()
    assert abs(syn.model_checksum(model) - float(g['model_checksum'])) < 1e-6 * float(g['model_checksum'])
    return g, d, model


@pytest.mark.parametrize('name', CASES)
def test_restatements_equal_the_references_init_guess(name):
    g, d, model = _load()
    views = list(g[name + '/views'])
    fs = float(g[name + '/fixed_scale'])
    est = fs < 0
    s0 = 1.0 if est else fs
    orc = cn.ClosureOracle(model, np.float64)
    z = dict(betas=np.zeros(10), global_orient=np.zeros(3), body_pose=np.zeros(69), transl=np.zeros(3), scale=np.array([s0]),
             use_vposer=False)
    rest = orc.body(z, want_cache=False)['joints']                        # init_guess.py:31-52
    kp = d['keypoints'].reshape(6, 17, 3).astype(np.float64)
    if len(views) == 1:
        j3 = ign.single_view_joints3d(rest, d['extris'][views[0]], d['intris'][views[0]], kp[views[0]])
    else:
        j3 = tn.recompute3d(d['extris'][views], d['intris'][views], kp[views])
    rot, trans, scale, _ = un.umeyama(rest[TORSO], j3[TORSO], est)
    assert np.abs(un.rotvec(rot) - g[name + '/global_orient']).max() < 1e-9
    assert np.abs(trans - g[name + '/transl']).max() < 1e-9 * max(1.0, np.abs(trans).max())
    assert abs((scale if est else s0) - float(g[name + '/scale'])) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_device_init_guess_equals_the_references(name):
    from mvsmplfitting_amd import init_guess as ig
    from tests.gpu_helpers import make_engine
    g, d, model = _load()
    views = list(g[name + '/views'])
    fs = float(g[name + '/fixed_scale'])
    est = fs < 0
    eng = make_engine(model)
    cams = tuple(d[k][views].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    kps = d['keypoints'].reshape(6, 17, 3)[views][None].astype(np.float32)          # [1, V, 17, 3]
    eng.set_problems(cams, kps[..., :2], kps[..., 2])
    out = ig.init_guess_batch(eng, d['extris'][views], d['intris'][views], kps, est_scale=est, fixed_scale=None if est else fs,
                              use_torso=True)
    go, tr, sc = (out[k][0].cpu().numpy() for k in ('global_orient', 'transl', 'scale'))
    assert np.abs(go - g[name + '/global_orient']).max() < 1e-4, (go, g[name + '/global_orient'])
    assert np.abs(tr - g[name + '/transl']).max() < 1e-4 * max(1.0, np.abs(g[name + '/transl']).max()), (tr, g[name + '/transl'])
    assert abs(float(sc) - float(g[name + '/scale'])) < 1e-4 * float(g[name + '/scale'])
    eng.close()
