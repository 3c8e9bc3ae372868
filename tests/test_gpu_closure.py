"""GPU parity: the HIP closure (through the C ABI) against the reference's golden vectors and
the NumPy oracle.  Tolerances are north_star's: vertices 1e-4 abs, scalar loss 1e-5 relative
(the loss tolerance can only be relative at float32)."""
import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from tests.gpu_helpers import flags_for, from118, make_engine, to118
from tests.helpers import CASES, body_model, load_case, oracle_for, stage_weights

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
VERT_ATOL = 1e-4
GRAD_RTOL = 2e-4        # relative to max |grad| (reference fp32-vs-fp64 itself: 2e-7..1e-5)


@pytest.mark.parametrize('sparse', [False, True])
@pytest.mark.parametrize('name', sorted(CASES))
def test_closure_matches_reference_golden(name, sparse):
    cfg, g, model, vpw, gmm, wts, cams = load_case(name)
    eng = make_engine(model, vpw, gmm)
    B = g['x'].shape[0]
    eng.set_problems(cams, g['gt_xy'], g['conf'])
    if 'joints3d' in g:
        eng.set_joints3d(g['joints3d'][:, :, :3], g['joints3d'][:, :, 3])
    x = np.stack([to118(g['x'][b], cfg['use_vposer']) for b in range(B)]).astype(np.float32)
    w = dict(wts)
    w['flags'] = flags_for(cfg) | (_lib.F_SPARSE_VERTS if sparse else 0)
    out = eng.closure(x, w, want_grad=True, want_verts=True, want_joints=True)
    loss = out['loss'].cpu().numpy().astype(np.float64)
    grad = out['grad'].cpu().numpy().astype(np.float64)
    verts = out['verts'].cpu().numpy().astype(np.float64)
    joints = out['joints'].cpu().numpy().astype(np.float64)
    assert np.all(np.abs(loss - g['loss64']) <= LOSS_RTOL * np.abs(g['loss64'])), (loss, g['loss64'])
    assert np.abs(joints - g['joints64']).max() < VERT_ATOL
    nvv = g['verts64_as32'].shape[0]
    assert np.abs(verts[:nvv] - g['verts64_as32']).max() < VERT_ATOL
    for b in range(B):
        gm = from118(grad[b], cfg['use_vposer'])
        gr = g['grad64'][b]
        if cfg.get('fix_shape'):
            assert np.all(gm[:10] == 0.0)
            gm = gm[10:]
        assert np.abs(gm - gr).max() <= GRAD_RTOL * np.abs(gr).max(), (name, b, np.abs(gm - gr).max(), np.abs(gr).max())
    # how close are we to the reference's own float32 run? (report, and a loose bound)
    e_mine = np.abs(loss - g['loss64']) / np.abs(g['loss64'])
    e_ref32 = np.abs(g['loss32'] - g['loss64']) / np.abs(g['loss64'])
    assert e_mine.max() <= max(20 * e_ref32.max(), 2e-6)
    eng.close()


def test_closure_batch32_against_oracle():
    """BASELINE config 2 shape: 32 frames x 8 views; loss/grad/vertices vs the float64 oracle."""
    model = body_model()
    cams = syn.make_camera_ring(8)
    orc = oracle_for(model, None, None)
    B = 32
    fr = syn.make_frames(B, seed0=1000)
    kps = []
    for b in range(B):
        p = {k: fr[k][b] for k in fr}
        p['use_vposer'] = False
        kps.append(orc.body(p, want_cache=False)['joints'])
    gt, conf = syn.make_observations(np.asarray(kps), cams, seed=99)
    rng = np.random.default_rng(17)
    x86 = rng.normal(0, 0.15, (B, 86))
    x86[:, 85] = 1.0 + rng.normal(0, 0.05, B)
    x = np.stack([to118(x86[b], False) for b in range(B)]).astype(np.float32)
    wts = stage_weights(2)
    eng = make_engine(model)
    eng.set_problems(cams, gt, conf)
    ref = [orc.closure(x[b, :86].astype(np.float64), cams, gt[b], conf[b], wts) for b in range(B)]      # float64 oracle, once
    for sparse in (False, True):
        w = dict(wts, flags=_lib.F_SPARSE_VERTS if sparse else 0)
        out = eng.closure(x, w, want_verts=True, want_joints=True)
        loss = out['loss'].cpu().numpy().astype(np.float64)
        grad = out['grad'].cpu().numpy().astype(np.float64)
        verts = out['verts'].cpu().numpy().astype(np.float64)
        for b in range(B):                                # all 32 problems of BASELINE configs[1]
            L, gq, o = ref[b]
            assert abs(loss[b] - L) <= LOSS_RTOL * abs(L), (b, sparse)
            assert np.abs(grad[b, :86] - gq).max() <= GRAD_RTOL * np.abs(gq).max(), (b, sparse)
            assert np.abs(verts[b] - o['vertices']).max() < VERT_ATOL, (b, sparse)
            assert np.all(grad[b, 86:] == 0.0)
    eng.close()


def test_closure_properties_full_size():
    """Size-independent properties at B = 32, V = 8."""
    model = body_model()
    cams = syn.make_camera_ring(8)
    B = 32
    rng = np.random.default_rng(3)
    x = np.zeros((B, 118), np.float32)
    x[:, :86] = rng.normal(0, 0.1, (B, 86))
    x[:, 85] = 1.0
    gt = rng.uniform(200, 1800, (B, 8, 17, 2)).astype(np.float32)
    conf = rng.uniform(0.3, 1.0, (B, 8, 17)).astype(np.float32)
    wts = dict(stage_weights(1), flags=0)
    eng = make_engine(model)
    eng.set_problems(cams, gt, conf)
    base = eng.closure(x, wts, want_verts=True)
    l0 = base['loss'].cpu().numpy()
    v0 = base['verts'].cpu().numpy()
    # (1) problems are independent: permuting the batch permutes the outputs bit-exactly
    perm = rng.permutation(B)
    eng.set_problems(cams, gt[perm], conf[perm])
    pl = eng.closure(x[perm], wts, want_verts=True)
    assert np.array_equal(pl['loss'].cpu().numpy(), l0[perm])
    assert np.array_equal(pl['verts'].cpu().numpy(), v0[perm])
    assert np.array_equal(pl['grad'].cpu().numpy(), base['grad'].cpu().numpy()[perm])
    # (2) translating a body translates every vertex (body_models_scale.py:401-403)
    x2 = x.copy()
    x2[:, 82:85] += np.array([0.25, -0.5, 0.125], np.float32)
    eng.set_problems(cams, gt, conf)
    v2 = eng.closure(x2, wts, want_verts=True)['verts'].cpu().numpy()
    assert np.abs((v2 - v0) - np.array([0.25, -0.5, 0.125])).max() < 2e-6
    # (3) zero confidence everywhere leaves only the priors; their gradient has no transl/scale part
    eng.set_problems(cams, gt, np.zeros_like(conf))
    pz = eng.closure(x, wts)
    gz = pz['grad'].cpu().numpy()
    assert np.all(gz[:, 82:86] == 0.0) and np.all(gz[:, 10:13] == 0.0)
    wp, ws = wts['body_pose_weight'], wts['shape_weight']
    bp = x[:, 13:82].astype(np.float64)
    P = (bp ** 2).sum(1) * wp ** 2
    P = np.where(P.astype(np.float32) > 5e4, 0.0, P)                 # fitting.py:334-335
    expect = P + (bp ** 2).sum(1) * (4 * wp) ** 2 + (x[:, :10].astype(np.float64) ** 2).sum(1) * ws ** 2
    ang = np.exp(bp[:, [52, 55, 9, 12]] * np.array([1, -1, -1, -1.0])) ** 2
    expect = expect + ang.sum(1) * wts['bending_prior_weight']
    assert np.all(np.abs(pz['loss'].cpu().numpy() - expect) <= 1e-5 * expect)
    # (4) dropping a view by zero confidence == removing it from the rig (main.py:49-57)
    conf_d = conf.copy()
    conf_d[:, 3] = 0.0
    eng.set_problems(cams, gt, conf_d)
    la = eng.closure(x, wts)['loss'].cpu().numpy()
    keep = [0, 1, 2, 4, 5, 6, 7]
    cams7 = tuple(c[keep] for c in cams)
    eng.set_problems(cams7, gt[:, keep], conf[:, keep])
    lb = eng.closure(x, wts)['loss'].cpu().numpy()
    assert np.all(np.abs(la - lb) <= 2e-6 * np.abs(lb))
    # (5) full-vertex mode and objective-vertices-only mode agree
    eng.set_problems(cams, gt, conf)
    ls = eng.closure(x, dict(wts, flags=_lib.F_SPARSE_VERTS))
    assert np.all(np.abs(ls['loss'].cpu().numpy() - l0) <= 2e-6 * np.abs(l0))
    gd = np.abs(ls['grad'].cpu().numpy() - base['grad'].cpu().numpy()).max()
    assert gd <= 1e-4 * np.abs(base['grad'].cpu().numpy()).max()
    eng.close()


def test_ragged_batches_and_errors():
    model = body_model()
    cams = syn.make_camera_ring(3)
    eng = make_engine(model)
    rng = np.random.default_rng(5)
    for B in (1, 31, 33):
        x = np.zeros((B, 118), np.float32)
        x[:, :86] = rng.normal(0, 0.1, (B, 86)); x[:, 85] = 1.0
        gt = rng.uniform(200, 1800, (B, 3, 17, 2)).astype(np.float32)
        conf = np.ones((B, 3, 17), np.float32)
        eng.set_problems(cams, gt, conf)
        o1 = eng.closure(x, dict(stage_weights(0), flags=0), want_verts=True)
        # same problems, evaluated alone, give the same bits
        eng.set_problems(cams, gt[B - 1:], conf[B - 1:])
        o2 = eng.closure(x[B - 1:], dict(stage_weights(0), flags=0), want_verts=True)
        assert np.array_equal(o1['loss'].cpu().numpy()[B - 1:], o2['loss'].cpu().numpy())
        assert np.array_equal(o1['verts'].cpu().numpy()[B - 1:], o2['verts'].cpu().numpy())
    from mvsmplfitting_amd.engine import MvFitError
    with pytest.raises(MvFitError):
        eng.closure(x[B - 1:], dict(stage_weights(0), flags=_lib.F_VPOSER))     # no decoder loaded
    with pytest.raises(MvFitError):
        eng.closure(x[B - 1:], dict(stage_weights(0), flags=0, coll_loss_weight=10.0))
    eng.close()


@pytest.mark.parametrize('V', [1, 16])
def test_view_count_extremes_and_per_problem_cameras(V):
    """1 view and MVFIT_MAX_VIEWS views, cameras given per problem ([B,V,...]) - against the oracle; one view
    more than the maximum is refused."""
    model = body_model()
    orc = oracle_for(model, None, None)
    B = 3
    rng = np.random.default_rng(40 + V)
    rigs = [syn.make_camera_ring(V, radius=3.5 + 0.4 * b, height=0.2 * b) for b in range(B)]
    cams_b = tuple(np.stack([rigs[b][i] for b in range(B)]) for i in range(4))      # [B,V,...]
    x86 = rng.normal(0, 0.12, (B, 86)); x86[:, 85] = 1.0 + rng.normal(0, 0.05, B)
    x = np.stack([to118(x86[b], False) for b in range(B)]).astype(np.float32)
    gt = rng.uniform(300, 1700, (B, V, 17, 2)).astype(np.float32)
    conf = rng.uniform(0.0, 1.0, (B, V, 17)).astype(np.float32)
    conf[conf < 0.2] = 0.0                                         # undetected keypoints
    wts = stage_weights(2)
    eng = make_engine(model)
    eng.set_problems(cams_b, gt, conf)
    for sparse in (False, True):
        out = eng.closure(x, dict(wts, flags=_lib.F_SPARSE_VERTS if sparse else 0))
        loss = out['loss'].cpu().numpy().astype(np.float64)
        grad = out['grad'].cpu().numpy().astype(np.float64)
        for b in range(B):
            Lr, gr, _ = orc.closure(x86[b].astype(np.float32).astype(np.float64), rigs[b], gt[b], conf[b], wts)
            assert abs(loss[b] - Lr) <= LOSS_RTOL * abs(Lr)
            assert np.abs(grad[b][:86] - gr).max() <= GRAD_RTOL * np.abs(gr).max()
    if V == 16:
        from mvsmplfitting_amd.engine import MvFitError
        cams17 = syn.make_camera_ring(17)
        with pytest.raises(MvFitError):
            eng.set_problems(cams17, np.zeros((1, 17, 17, 2), np.float32), np.ones((1, 17, 17), np.float32))
    eng.close()


def test_sparse_skinning_path_is_bit_identical_to_dense(monkeypatch):
    """A model with <= 4 weights per vertex (like SMPL) takes the 4-pair blend in the vertex pass; forcing the
    dense 24-column blend on the same model must give the same bits (same non-zero products, same order)."""
    model = body_model(0, 4)
    cams = syn.make_camera_ring(8)
    B = 33
    rng = np.random.default_rng(8)
    x = np.zeros((B, 118), np.float32)
    x[:, :86] = rng.normal(0, 0.2, (B, 86)); x[:, 85] = 1.0 + rng.normal(0, 0.05, B)
    gt = rng.uniform(300, 1700, (B, 8, 17, 2)).astype(np.float32)
    conf = np.ones((B, 8, 17), np.float32)
    outs = []
    for dense in (0, 1):
        eng = make_engine(model, dense_skinning=dense)
        eng.set_problems(cams, gt, conf)
        o = eng.closure(x, dict(stage_weights(1), flags=0), want_verts=True)
        outs.append((o['verts'].cpu().numpy(), o['loss'].cpu().numpy(), o['grad'].cpu().numpy()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


def test_split_fp16_contraction_against_exact_fp32(monkeypatch):
    """The default vertex pass takes the blendshape contraction as error-compensated split-fp16 products on the
    fp16 matrix pipe; contraction = 'exact_fp32' keeps the exact fp32 MFMA chain.  The two agree to fp32 rounding level
    on the vertices (both are ~5e-7 from the float64 oracle), far inside the 1e-4 tolerance."""
    model = body_model()
    cams = syn.make_camera_ring(8)
    orc = oracle_for(model, None, None)
    B = 32
    rng = np.random.default_rng(21)
    x86 = rng.normal(0, 0.3, (B, 86)); x86[:, :10] = rng.normal(0, 1.5, (B, 10)); x86[:, 85] = 1.0 + rng.normal(0, 0.05, B)
    x = np.stack([to118(x86[b], False) for b in range(B)]).astype(np.float32)
    gt = rng.uniform(300, 1700, (B, 8, 17, 2)).astype(np.float32)
    conf = np.ones((B, 8, 17), np.float32)
    outs = {}
    for exact in ('0', '1'):
        eng = make_engine(model, contraction='exact_fp32' if exact == '1' else 'split_fp16')
        eng.set_problems(cams, gt, conf)
        o = eng.closure(x, dict(stage_weights(2), flags=0), want_verts=True)
        outs[exact] = (o['verts'].cpu().numpy().astype(np.float64), o['loss'].cpu().numpy().astype(np.float64))
        eng.close()
    assert np.abs(outs['0'][0] - outs['1'][0]).max() < 2e-6
    assert np.all(np.abs(outs['0'][1] - outs['1'][1]) <= 2e-6 * np.abs(outs['1'][1]))
    errs = {}
    for b in (0, 7, 31):
        ref = orc.body(dict(cn.unpack(x[b, :86].astype(np.float64), False), use_vposer=False), want_cache=False)['vertices']
        for k in outs:
            errs.setdefault(k, []).append(np.abs(outs[k][0][b] - ref).max())
    assert max(errs['0']) < 3e-6 and max(errs['0']) <= 4 * max(errs['1']) + 1e-6, errs


def test_half_width_basis_operands(monkeypatch):
    """BASELINE configs[4] ("bf16 LBS with MFMA shapedirs contraction"): contraction = 'half_basis' streams the blendshape basis
    at 2 bytes per element (the fp16 hi halves of the split operands: 11 significant bits where bf16 has 8).  Relaxed,
    stated tolerance: vertices within 1e-4 of the float64 oracle (SURVEY 8(d) config 5 expected <~ 1e-4); everything
    else of the closure is unchanged (the objective's own 69 vertices do not go through the pass in the fit)."""
    model = body_model(0, 4)
    orc = oracle_for(model, None, None)
    rng = np.random.default_rng(23)
    B = 32
    x = np.zeros((B, 118), np.float32)
    x[:, :86] = rng.normal(0, 0.3, (B, 86))
    x[:, 85] = 1.0
    eng = make_engine(model, contraction='half_basis')
    ref = make_engine(model)
    cams = syn.make_camera_ring(8)
    for e in (eng, ref):
        e.set_problems(cams, np.zeros((B, 8, 17, 2), np.float32), np.ones((B, 8, 17), np.float32))
    vh, _ = eng.vertices(x)
    vf, _ = ref.vertices(x)
    vh, vf = vh.cpu().numpy().astype(np.float64), vf.cpu().numpy().astype(np.float64)
    worst = 0.0
    for b in range(0, B, 7):
        p = dict(betas=x[b, :10], global_orient=x[b, 10:13], body_pose=x[b, 13:82], transl=x[b, 82:85], scale=x[b, 85:86],
                 use_vposer=False)
        o = orc.body({k: (np.asarray(v, np.float64) if k != 'use_vposer' else v) for k, v in p.items()}, want_cache=False)
        assert np.abs(vf[b] - o['vertices']).max() < 2e-6
        worst = max(worst, np.abs(vh[b] - o['vertices']).max())
    assert 1e-7 < worst < 1e-4, worst                         # really the half-width path, inside the relaxed tolerance
    eng.close(); ref.close()
