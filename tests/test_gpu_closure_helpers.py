"""GPU parity of the VPoser decoder the production fits actually use (code/model/VPoser.py:218-232): with
mvfit_options::closure_vposer_helpers = 1: mvfit_closure decodes the body pose on the helper workgroups of its own launch - the
register-stationary decoder of the single-launch fits (csrc/vposer_service.h), whose summation order differs from the
in-workgroup decoder - and, like those fits, evaluates the objective from the vertices it computes itself while the full
vertex pass runs on the operands it published.  Same goldens (the reference's own float64 closure), same tolerances as
tests/test_gpu_closure.py / test_gpu_demo.py: loss 1e-5 relative, gradient 2e-4 of its maximum, vertices 1e-4."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from tests.gpu_helpers import flags_for, from118, make_engine, to118
from tests.helpers import GOLD, body_model, load_case

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
VERT_ATOL = 1e-4
GRAD_RTOL = 2e-4


@pytest.fixture
def helper_route(monkeypatch):
    from mvsmplfitting_amd import engine
    monkeypatch.setitem(engine.DEFAULT_OPTIONS, 'closure_vposer_helpers', 1)     # every engine of the test: mvfit_closure decodes on helpers


@pytest.mark.parametrize('name', ['vp_s0_v8', 'vpwild_s2_v8'])
def test_closure_through_the_decoder_helpers_matches_reference_golden(name, helper_route):
    cfg, g, model, vpw, gmm, wts, cams = load_case(name)
    assert cfg['use_vposer']
    eng = make_engine(model, vpw, gmm)
    B = g['x'].shape[0]
    eng.set_problems(cams, g['gt_xy'], g['conf'])
    x = np.stack([to118(g['x'][b], True) for b in range(B)]).astype(np.float32)
    w = dict(wts, flags=flags_for(cfg))
    out = eng.closure(x, w, want_grad=True, want_verts=True, want_joints=True)
    dec = eng.decoder_stats()
    assert dec['launches'] == 1 and dec['answers_timed_out'] == 0 and dec['helpers_gave_up'] == 0, dec
    loss = out['loss'].cpu().numpy().astype(np.float64)
    grad = out['grad'].cpu().numpy().astype(np.float64)
    assert np.all(np.abs(loss - g['loss64']) <= LOSS_RTOL * np.abs(g['loss64'])), (loss, g['loss64'])
    assert np.abs(out['joints'].cpu().numpy() - g['joints64']).max() < VERT_ATOL
    nvv = g['verts64_as32'].shape[0]
    assert np.abs(out['verts'].cpu().numpy()[:nvv] - g['verts64_as32']).max() < VERT_ATOL
    for b in range(B):
        gm, gr = from118(grad[b], True), g['grad64'][b]
        assert np.abs(gm - gr).max() <= GRAD_RTOL * np.abs(gr).max(), (name, b, np.abs(gm - gr).max(), np.abs(gr).max())
    # next to the reference's own float32 run
    e_ref32 = np.abs(g['loss32'] - g['loss64']) / np.abs(g['loss64'])
    assert (np.abs(loss - g['loss64']) / np.abs(g['loss64'])).max() <= max(20 * e_ref32.max(), 2e-6)
    # and it IS another decoder than the in-workgroup one (the route is taken): same values to ~1e-6, not bit for bit
    eng.set_options(closure_vposer_helpers=0)
    loc = eng.closure(x, dict(w, flags=w['flags'] | _lib.F_SPARSE_VERTS), want_grad=True)
    l2 = loc['loss'].cpu().numpy().astype(np.float64)
    assert np.all(np.abs(l2 - loss) <= 2e-6 * np.abs(loss))
    eng.close()


def test_demo_closure_through_the_decoder_helpers(helper_route):
    """BASELINE configs[0]: the shipped checkpoint's decoder on the helpers, the demo's real cameras / keypoints."""
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    cams = tuple(g[k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    stages = [dict(data_weight=float(w[0]), body_pose_weight=float(w[1]), shape_weight=float(w[2]),
                   bending_prior_weight=float(w[3]), rho=float(w[4]), flags=_lib.F_VPOSER) for w in g['stage_w']]
    eng = make_engine(body_model(), vpw)
    n = g['cx'].shape[0]
    eng.set_problems(cams, np.repeat(g['gt_xy'][None], n, 0), np.repeat(g['conf'][None], n, 0))
    x = np.stack([to118(xx, True) for xx in g['cx']]).astype(np.float32)
    k = 0
    for si in (0, 3):
        out = eng.closure(x, dict(stages[si]), want_grad=True, want_verts=True, want_joints=True)
        dec = eng.decoder_stats()
        assert dec['launches'] == 1 and dec['answers_timed_out'] == 0, dec
        loss = out['loss'].cpu().numpy().astype(np.float64)
        grad = out['grad'].cpu().numpy().astype(np.float64)
        ref_l = g['closs64'][k:k + n]
        assert np.all(np.abs(loss - ref_l) <= LOSS_RTOL * np.abs(ref_l)), (si, loss, ref_l)
        assert np.abs(out['joints'].cpu().numpy() - g['cjoints64'][k:k + n]).max() < VERT_ATOL
        if si == 0:
            assert np.abs(out['verts'].cpu().numpy() - g['cverts64_as32']).max() < VERT_ATOL
        for b in range(n):
            gm, gr = from118(grad[b], True), g['cgrad64'][k + b]
            assert np.abs(gm - gr).max() <= GRAD_RTOL * np.abs(gr).max(), (si, b, np.abs(gm - gr).max(), np.abs(gr).max())
        k += n
    eng.close()
