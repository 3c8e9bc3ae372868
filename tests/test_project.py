"""Per-view projection of the full mesh (reference cam(verts), code/utils/utils.py:603-607):
CPU: the NumPy restatement against goldens written by the reference's own PerspectiveCamera; GPU: mvfit_project_points
against the same goldens (float pixels to 2e-3 px - the reference's own float32 run is that far from its float64
one - and identical integer pixels, which is what the reference draws, wherever the float64 value is not within that
distance of an integer)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import synthetic as syn
from oracle import project_np
from tests.helpers import GOLD

G = dict(np.load(os.path.join(GOLD, 'project.npz')))
IDX = G['idx']                     # the goldens hold the reference's pixels for every 8th point
PX_TOL = 2e-3


def test_numpy_restatement_equals_reference_camera():
    cams = tuple(G['demo_' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    uv = project_np.project(G['demo_pts'].astype(np.float64), cams)[:, IDX]
    assert np.abs(uv - G['demo_uv64']).max() < 1e-9
    ring = syn.make_camera_ring(8)
    for b in range(2):
        uv = project_np.project(G['ring_pts'][b].astype(np.float64), ring)[:, IDX]
        assert np.abs(uv - G['ring_uv64'][b]).max() < 1e-9
    assert np.abs(G['demo_uv32'] - G['demo_uv64']).max() < PX_TOL and np.abs(G['ring_uv32'] - G['ring_uv64']).max() < PX_TOL


def _check(uv, ref64):
    assert uv.shape == ref64.shape
    assert np.abs(uv - ref64).max() < PX_TOL, np.abs(uv - ref64).max()
    safe = np.abs(ref64 - np.round(ref64)) > PX_TOL                 # truncation is stable there
    assert safe.mean() > 0.99
    assert np.array_equal(uv.astype(np.int32)[safe], ref64.astype(np.int32)[safe])      # utils.py:604-605 .astype(np.int32)


@pytest.mark.gpu
def test_gpu_projection_matches_reference_camera():
    from tests.gpu_helpers import make_engine
    from tests.helpers import body_model
    eng = make_engine(body_model())
    cams = tuple(G['demo_' + k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    eng.set_problems(cams, np.zeros((1, 6, 17, 2), np.float32), np.ones((1, 6, 17), np.float32))
    _check(eng.project(G['demo_pts'][None]).cpu().numpy()[0][:, IDX], G['demo_uv64'])
    ring = syn.make_camera_ring(8)
    eng.set_problems(ring, np.zeros((2, 8, 17, 2), np.float32), np.ones((2, 8, 17), np.float32))
    uv = eng.project(G['ring_pts']).cpu().numpy()
    for b in range(2):
        _check(uv[b][:, IDX], G['ring_uv64'][b])
    # per-problem cameras (cam_batched) and a ragged point count
    Rb = np.stack([ring[0], ring[0][::-1]]); tb = np.stack([ring[1], ring[1][::-1]])
    fb = np.stack([ring[2], ring[2][::-1]]); cb = np.stack([ring[3], ring[3][::-1]])
    eng.set_problems((Rb, tb, fb, cb), np.zeros((2, 8, 17, 2), np.float32), np.ones((2, 8, 17), np.float32))
    uv2 = eng.project(G['ring_pts'][:, :1001]).cpu().numpy()
    k = IDX[IDX < 1001]
    assert np.abs(uv2[0][:, k] - G['ring_uv64'][0][:, :len(k)]).max() < PX_TOL
    assert np.abs(uv2[1][:, k] - G['ring_uv64'][1][::-1, :len(k)]).max() < PX_TOL
    # the mesh of a fit result, projected like visualize_fitting does: vertices -> every view
    verts, joints = eng.vertices(np.tile(np.eye(1, 118, 85, dtype=np.float32), (2, 1)))
    uvv = eng.project(verts)
    assert uvv.shape == (2, 8, 6890, 2) and bool(np.isfinite(uvv.cpu().numpy()).all())
    eng.close()
