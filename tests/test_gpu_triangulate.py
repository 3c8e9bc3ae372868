"""GPU parity: batched multi-view triangulation (mvfit_triangulate) against golden vectors written by the reference's
own recompute3D (oracle/make_golden_triangulate.py), plus the mirror's reference-shaped call."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import init_guess as ig
from tests.gpu_helpers import make_engine
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(GOLD, 'triangulate.npz')))


@pytest.mark.parametrize('name', ['v8', 'v2', 'v16'])
def test_triangulation_matches_reference_golden(name):
    eng = make_engine(body_model())
    e, i, k, ref = G[name + '_extris'], G[name + '_intris'], G[name + '_kps'], G[name + '_joints3d']
    out = ig.recompute3D_batch(eng, e, i, k).cpu().numpy()
    # float64 arithmetic with the reference's float32 rounding of AtA (recompute3D.py:54): a last-bit float64
    # difference in an accumulated element can land on the other side of a float32 rounding boundary, i.e. move that
    # element by one float32 ulp (6e-8 relative) - the solutions then differ by ~1e-7 m; tolerance 1e-6 m
    # - scaled by the conditioning of the joint's 3x3 system: |dx| <= cond * ulp32 * |x| (with a factor 4 of slack)
    from oracle import triangulate_np as tn
    for b in range(k.shape[0]):
        _, AtA, _ = tn.recompute3d(e, i, k[b], return_system=True)
        cond = np.array([np.linalg.cond(AtA[j].astype(np.float64)) for j in range(17)])
        tol = 4 * 6e-8 * cond * np.maximum(1.0, np.abs(ref[b]).max(1)) + 1e-9
        err = np.abs(out[b] - ref[b]).max(1)
        assert np.all(err <= tol), (b, err.max(), tol[err.argmax()], cond.max())
    # reference-shaped call for one frame: lists over views of [1,17,3]
    one = ig.recompute3D(eng, list(e), list(i), [k[0, v][None] for v in range(k.shape[1])])
    assert np.abs(one - out[0]).max() == 0.0
    eng.close()


def test_triangulation_properties():
    """Noise-free observations triangulate to the generating points; a joint seen by no view still solves (the
    reference's 1e-6 floor on the confidence keeps the normal equations regular)."""
    from mvsmplfitting_amd import synthetic as syn
    eng = make_engine(body_model())
    V, B = 5, 40
    rng = np.random.default_rng(5)
    cam_R, cam_t, cam_f, cam_c = syn.make_camera_ring(V)
    extris = np.tile(np.eye(4), (V, 1, 1)); extris[:, :3, :3] = cam_R; extris[:, :3, 3] = cam_t
    intris = np.zeros((V, 3, 3)); intris[:, 0, 0] = cam_f; intris[:, 1, 1] = cam_f; intris[:, 0, 2] = cam_c[:, 0]
    intris[:, 1, 2] = cam_c[:, 1]; intris[:, 2, 2] = 1
    X = rng.normal(0, 0.5, (B, 17, 3))
    kps = np.zeros((B, V, 17, 3), np.float32)
    for v in range(V):
        p = X @ extris[v, :3, :3].T + extris[v, :3, 3]
        uv = p @ intris[v].T
        kps[:, v, :, :2] = uv[..., :2] / uv[..., 2:3]
        kps[:, v, :, 2] = 1.0
    out = eng.triangulate(kps, intris, extris).cpu().numpy()
    assert np.abs(out - X).max() < 2e-3                     # float32 pixel coordinates at f = 2400
    kps[:, :, 3, 2] = 0.0                                   # joint 3 unseen everywhere: still finite
    out2 = eng.triangulate(kps, intris, extris).cpu().numpy()
    assert np.all(np.isfinite(out2)) and np.abs(out2[:, 3] - X[:, 3]).max() < 2e-3
    from mvsmplfitting_amd.engine import MvFitError
    with pytest.raises(MvFitError):
        eng.triangulate(kps, intris[:3], extris)
    eng.close()
