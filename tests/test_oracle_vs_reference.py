"""CPU, build container only: the oracle against the live reference (skips on the GPU box)."""
import numpy as np
import pytest

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import ref_import as ri
from tests.helpers import body_model, stage_weights

pytestmark = pytest.mark.skipif(not ri.available(), reason='reference tree not mounted')


@pytest.mark.parametrize('use_vp', [False, True])
def test_live_reference_fp64(use_vp):
    model = body_model()
    cams = syn.make_camera_ring(5)
    vpw = syn.make_vposer_decoder(seed=3, gain=2.0, identity_bias=False) if use_vp else None
    orc = cn.ClosureOracle(model, np.float64, vposer=vpw)
    fr = syn.make_frames(1, seed0=31)
    p = {k: fr[k][0] for k in fr}
    p['use_vposer'] = False
    kp = orc.body(p, want_cache=False)['joints']
    gt, cf = syn.make_observations(kp[None], cams, seed=1)
    lay, D = cn.param_layout(use_vp)
    x = np.random.default_rng(5).normal(0, 0.2, D)
    x[lay['scale'][0]] = 0.95
    wts = stage_weights(2)
    L, g, out = orc.closure(x, cams, gt[0], cf[0], wts, use_vposer=use_vp)
    rp = ri.RefProblem(model, cams, gt[0], cf[0], 'float64', use_vposer=use_vp, vposer_weights=vpw)
    Lr, gr, vr, jr = rp.eval_closure(x, wts)
    assert abs(L - Lr) <= 1e-13 * abs(Lr)
    assert np.abs(g - gr).max() <= 1e-11 * np.abs(gr).max()
    assert np.abs(out['vertices'] - vr).max() < 1e-13
    assert np.abs(out['joints'] - jr).max() < 1e-13


def test_live_reference_3d_joint_term():
    """use_3d (fitting.py:319-324): oracle vs the reference with gt_joints3d / joints3d_conf."""
    model = body_model()
    cams = syn.make_camera_ring(4)
    orc = cn.ClosureOracle(model, np.float64)
    fr = syn.make_frames(1, seed0=77)
    p = {k: fr[k][0] for k in fr}
    p['use_vposer'] = False
    kp = orc.body(p, want_cache=False)['joints']
    gt, cf = syn.make_observations(kp[None], cams, seed=2)
    rng = np.random.default_rng(9)
    j3 = (kp + rng.normal(0, 0.05, kp.shape), rng.uniform(0.2, 1.0, 17))
    lay, D = cn.param_layout(False)
    x = rng.normal(0, 0.2, D)
    x[lay['scale'][0]] = 1.05
    wts = stage_weights(1)
    L, g, out = orc.closure(x, cams, gt[0], cf[0], wts, joints3d=j3)
    rp = ri.RefProblem(model, cams, gt[0], cf[0], 'float64', joints3d=j3)
    Lr, gr, vr, jr = rp.eval_closure(x, wts)
    assert abs(L - Lr) <= 1e-13 * abs(Lr)
    assert np.abs(g - gr).max() <= 1e-11 * np.abs(gr).max()
    L0, _, _ = orc.closure(x, cams, gt[0], cf[0], wts)
    assert L > L0          # the term is really there


def test_triangulation_oracle_equals_reference_recompute3D():
    """oracle/triangulate_np.py against the reference's own function on the seeded rigs of the golden file."""
    from oracle import make_golden_triangulate as mg, triangulate_np as tn
    ri.load()
    from utils.recompute3D import recompute3D
    for V, B, seed in [(8, 2, 11), (3, 2, 12)]:
        extris, intris, kps = mg.make_case(V, B, seed)
        for b in range(B):
            ref = recompute3D(list(extris), list(intris), [kps[b, v][None].copy() for v in range(V)])
            assert np.abs(tn.recompute3d(extris, intris, kps[b]) - ref).max() <= 1e-12



def test_oracle_full_pose_against_reference_golden():
    """The NumPy restatement's VPoser decode + full_pose against ModelOutput.full_pose recorded from the reference
    (tests/golden/full_pose.npz; oracle/make_golden_full_pose.py)."""
    import os
    from mvsmplfitting_amd import synthetic as syn
    from tests.helpers import GOLD, body_model, oracle_for
    g = dict(np.load(os.path.join(GOLD, 'full_pose.npz')))
    real = dict(np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')))
    real = {k: real[k] for k in ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'out_w', 'out_b')}
    for name, vpw in (('real', real), ('wild', syn.make_vposer_decoder(seed=2, gain=1.0, identity_bias=False))):
        orc = oracle_for(body_model(0, 4), vpw, None)
        for x, ref in zip(g[name + '_x'], g[name + '_full_pose']):
            out = orc.body(dict(betas=x[0:10], global_orient=x[10:13], transl=x[13:16], scale=x[16], pose_embedding=x[17:49]),
                           want_cache=False)
            assert np.abs(out['full_pose'] - ref).max() < 1e-12
