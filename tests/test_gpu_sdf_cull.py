"""GPU: the all-faces interpenetration term on face lists (sdf_term.hip: projective bins for the crossing parity, cells for
the minimum distance) must give the brute-force kernel's bits - the walk over every face for every corner that the
reference performs (sdf/sdf/csrc/sdf_cuda_kernel.cu:258-287).  mvfit_options::sdf_face_lists = 0 keeps the brute-force kernel; both
engines see the same posed bodies.  The brute-force kernel itself is pinned against the reference's kernel source in
tests/test_gpu_sdf.py (same per-voxel code, sdf_device.h) and against the oracle in tests/test_gpu_sdf_term.py."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights

pytestmark = pytest.mark.gpu


def _term(model, x, cams, gt, conf, num_faces, G, cull):
    eng = MvFit(model, options=dict(sdf_face_lists=1 if cull else 0))
    eng.set_problems(cams, gt, conf)
    eng.set_sdf(model['faces'], num_faces=num_faces, grid_size=G)
    w = dict(stage_weights(1536.0, coll_w=[0.0, 0.0, 1000.0, 4500.0])[3])
    out = eng.closure(x, w, want_grad=True)
    smp, S = eng.sdf_term_read()
    res = (smp.cpu().numpy(), S.cpu().numpy(), out['loss'].cpu().numpy(), out['grad'].cpu().numpy())
    eng.close()
    return res


def _poses(B, seed, spread):
    fr = syn.make_frames(B, seed0=seed)
    x = np.zeros((B, 118), np.float32)
    for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items():
        x[:, a:b] = fr[k]
    rng = np.random.default_rng(seed)
    x[:, 13:82] += rng.normal(0, spread, (B, 69)).astype(np.float32)          # folded limbs: self-contact, thin and crossing triangles
    return x


@pytest.mark.parametrize('num_faces,G,spread', [(None, 128, 0.0), (None, 128, 0.6), (None, 32, 0.3), (3000, 128, 0.3), (None, 7, 0.3)])
def test_face_lists_give_the_bits_of_the_walk_over_all_faces(num_faces, G, spread):
    B, V = 6, 4
    model = syn.make_body_model(0, skin_topk=4)
    cams = syn.make_camera_ring(V)
    x = _poses(B, 4100 + G, spread)
    gt = np.zeros((B, V, 17, 2), np.float32)
    conf = np.ones((B, V, 17), np.float32)
    if spread == 0.0:
        # fits that ran off (bench.py --config configs2 --sdf-faces all does, as the walk over all faces did): astronomically
        # large but finite parameters, an infinite translation, a NaN - the box of such a body is not finite
        x[1, 13:82] *= 1e9
        x[2, 82] = np.inf
        x[3, 20] = np.nan
        x[4, 0:10] = 3e37
    a = _term(model, x, cams, gt, conf, num_faces, G, cull=True)
    b = _term(model, x, cams, gt, conf, num_faces, G, cull=False)
    inside = (np.nan_to_num(b[0][..., 0]) != 0).sum()
    assert inside > 100 or G < 16, 'the case does not exercise the term'
    for u, v, name in zip(a, b, ('samples', 'S', 'loss', 'grad')):
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32)), name


def test_a_staged_fit_with_all_faces_is_the_same_fit_on_lists_and_by_the_walk():
    """The whole staged fit (two leading stages asynchronous, the stages with the term chained: pass -> box -> lists ->
    samples -> entries -> pull-back -> step) lands on the same parameters, bit for bit, whichever kernel samples."""
    B, V, G = 3, 4, 32
    model = syn.make_body_model(0, skin_topk=4)
    cams = syn.make_camera_ring(V)
    xgt = _poses(B, 4242, 0.2)
    res = []
    for cull in (True, False):
        eng = MvFit(model, options=dict(sdf_face_lists=1 if cull else 0))
        eng.set_problems(cams, np.zeros((B, V, 17, 2), np.float32), np.ones((B, V, 17), np.float32))
        _, joints = eng.vertices(xgt)
        gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=77)
        eng.set_problems(cams, gt, conf)
        eng.set_sdf(model['faces'], num_faces=None, grid_size=G)
        x0 = np.zeros((B, 118), np.float32)
        x0[:, 85] = 1
        stages = stage_weights(1536.0, coll_w=[0.0, 0.0, 0.01, 0.05])       # small weights: the fit stays a fit
        xf, st = eng.fit(x0, stages)
        assert eng.sdf_info()['term'] == ('face_lists' if cull else 'walk'), eng.sdf_info()
        res.append((xf.cpu().numpy(), st['final_loss'].cpu().numpy(), st['n_closure'].cpu().numpy()))
        eng.close()
    assert np.isfinite(res[0][1]).all()
    assert np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
