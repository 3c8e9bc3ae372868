"""GPU: BASELINE configs[0] - the reference's shipped demo (cfg_files/fit_smpl.yaml) on its REAL inputs: the six
calibrated cameras of data/3DOH50K_Parameters.txt, the six keypoint files of data/keypoints/0000, image height 1536,
use_vposer with the decoder of the shipped checkpoint priors/snapshots/poser_epoch091.pkl, the yaml's weights and
optimiser settings; the body is the seeded synthetic one (no SMPL file ships).  Golden values were produced by the
reference's own code in the build container (oracle/make_golden_demo.py -> tests/golden/demo_fit_smpl.npz,
vposer_poser_epoch091_decoder.npz)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from tests.gpu_helpers import from118, make_engine, to118
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu


def _load():
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    model = body_model()
    cams = (g['cam_R'].astype(np.float32), g['cam_t'].astype(np.float32), g['cam_f'].astype(np.float32),
            g['cam_c'].astype(np.float32))
    stages = [dict(data_weight=float(w[0]), body_pose_weight=float(w[1]), shape_weight=float(w[2]),
                   bending_prior_weight=float(w[3]), rho=float(w[4]), flags=_lib.F_VPOSER) for w in g['stage_w']]
    return g, vpw, model, cams, stages


def test_demo_closure_with_the_shipped_vposer_checkpoint():
    g, vpw, model, cams, stages = _load()
    assert cams[0].shape == (6, 3, 3) and g['gt_xy'].shape == (6, 17, 2)
    eng = make_engine(model, vpw)
    n = g['cx'].shape[0]
    eng.set_problems(cams, np.repeat(g['gt_xy'][None], n, 0), np.repeat(g['conf'][None], n, 0))
    x = np.stack([to118(xx, True) for xx in g['cx']]).astype(np.float32)
    k = 0
    for si in (0, 3):
        for sparse in (False, True):
            w = dict(stages[si])
            w['flags'] |= _lib.F_SPARSE_VERTS if sparse else 0
            out = eng.closure(x, w, want_grad=True, want_verts=True, want_joints=True)
            loss = out['loss'].cpu().numpy().astype(np.float64)
            grad = out['grad'].cpu().numpy().astype(np.float64)
            ref_l = g['closs64'][k:k + n]
            assert np.all(np.abs(loss - ref_l) <= 1e-5 * np.abs(ref_l)), (si, loss, ref_l)
            assert np.abs(out['joints'].cpu().numpy() - g['cjoints64'][k:k + n]).max() < 1e-4
            if si == 0:
                assert np.abs(out['verts'].cpu().numpy() - g['cverts64_as32']).max() < 1e-4
            for b in range(n):
                gm, gr = from118(grad[b], True), g['cgrad64'][k + b]
                assert np.abs(gm - gr).max() <= 2e-4 * np.abs(gr).max(), (si, b, np.abs(gm - gr).max(), np.abs(gr).max())
            # next to the reference's own float32 run
            e_ref32 = np.abs(g['closs32'][k:k + n] - ref_l) / np.abs(ref_l)
            assert (np.abs(loss - ref_l) / np.abs(ref_l)).max() <= max(20 * e_ref32.max(), 2e-6)
        k += n
    eng.close()


@pytest.mark.parametrize('sparse', [False, True])
def test_demo_four_stage_fit(sparse):
    """The yaml's four stages from the reference's initial guess.  Which optimum is reached is chaotic in the last bits
    on this ill-conditioned problem: the reference's own float32 / float64 fits end at 36882 / 36490 and its float32
    fits from six starts perturbed by 1e-6 (relative) spread over 34257 ... 38929 after 580 ... 605 closures (all in the
    golden file).  The assertion is on the quality of the optimum and on the effort, with that spread as yard-stick."""
    g, vpw, model, cams, stages = _load()
    eng = make_engine(model, vpw)
    # five fits in one batch: the reference's start and four starts perturbed by 1e-6 (relative), as the reference's own
    # spread was recorded.  The reference's OWN float32 fits from 48 such starts (tests/golden/demo_spread48.npz, written by
    # oracle/make_golden_demo_spread.py) land in 34.2 k ... 39.9 k except for 2 of 48 that end at 44.4 k - the same second
    # optimum, at the same rate (~4 %), that the device fits show (test_demo_fit_spread_against_the_reference_spread below holds
    # the two distributions against each other).  Here: the MEDIAN of the five inside the six-start band, and no single fit
    # worse than the worst the reference itself reaches over its 48 starts.
    NB = 5
    eng.set_problems(cams, np.repeat(g['gt_xy'][None], NB, 0), np.repeat(g['conf'][None], NB, 0))
    x0 = np.repeat(to118(g['x0'], True)[None], NB, 0)
    rng = np.random.default_rng(20240)
    x0[1:] *= 1.0 + 1e-6 * rng.standard_normal(x0[1:].shape)
    x0 = x0.astype(np.float32)
    st_w = [dict(s, flags=s['flags'] | (_lib.F_SPARSE_VERTS if sparse else 0)) for s in stages]
    xf, st = eng.fit(x0, st_w)
    finals = st['final_loss'].cpu().numpy().astype(np.float64)
    ncls = st['n_closure'].cpu().numpy()
    final, ncl = float(np.median(finals)), int(np.median(ncls))
    ref_hi = max(float(g['fit_final32']), float(g['fit_final64']), float(g['fit_spread32'].max()))
    ref_n = [int(g['fit_ncl32'].sum()), int(g['fit_ncl64'].sum())] + [int(n) for n in g['fit_spread_ncl32'].sum(1)]
    ref_worst48 = float(np.load(os.path.join(GOLD, 'demo_spread48.npz'))['final32'].max())
    assert np.isfinite(finals).all() and final <= 1.02 * ref_hi and finals.max() <= 1.02 * ref_worst48, (finals, ref_hi, ref_worst48)
    assert 0.5 * min(ref_n) <= ncl <= 2.0 * max(ref_n), (ncls, ref_n)
    chk = eng.closure(xf, dict(st_w[-1]), want_grad=False)['loss'].cpu().numpy().astype(np.float64)
    assert (chk <= finals * (1 + 1e-3)).all()
    eng.close()


@pytest.mark.parametrize('sparse', [False, True])
def test_demo_fit_spread_against_the_reference_spread(sparse):
    """The SAME 192 starts (the reference's initial guess + 191 copies perturbed by 1e-6, relative) fitted by the reference itself
    in float32 (tests/golden/demo_spread192.npz, oracle/make_golden_demo_spread.py: 183 fits in 34.1 k ... 40.3 k, 9 at 44.4 k) and
    by the device in one batch.  Which optimum a start reaches is chaotic in the last bits, so the two are compared as
    distributions: the device's share of fits outside the main band must be compatible with the reference's - Fisher's exact
    test, TWO-sided, 5 % level (round 6: 192 starts instead of 48 and a one-sided 1 % level, which could not tell 4 % from 20 %) -,
    its median and its worst fit must not be worse than the reference's (2 %), and the effort must be comparable."""
    g, vpw, model, cams, stages = _load()
    sp = np.load(os.path.join(GOLD, 'demo_spread192.npz'))
    x0, ref = sp['x0'].astype(np.float32), sp['final32']
    n = x0.shape[0]
    eng = make_engine(model, vpw)
    eng.set_problems(cams, np.repeat(g['gt_xy'][None], n, 0), np.repeat(g['conf'][None], n, 0))
    st_w = [dict(s, flags=s['flags'] | (_lib.F_SPARSE_VERTS if sparse else 0)) for s in stages]
    xf, st = eng.fit(x0, st_w)
    dev = st['final_loss'].cpu().numpy().astype(np.float64)
    ncl = st['n_closure'].cpu().numpy()
    band = 1.1 * float(np.median(ref))                       # 40.9 k: between the main band and the second optimum
    out_ref, out_dev = int((ref > band).sum()), int((dev > band).sum())
    print('demo, %d starts: reference float32 median %.0f max %.0f, %d outside the band; device median %.0f max %.0f, %d outside; '
          'closures reference %d (median), device %d' % (n, np.median(ref), ref.max(), out_ref, np.median(dev), dev.max(), out_dev,
                                                       int(np.median(sp['ncl32'].sum(1))), int(np.median(ncl))))
    assert np.isfinite(dev).all()
    from scipy.stats import fisher_exact
    p_two = fisher_exact([[out_dev, n - out_dev], [out_ref, n - out_ref]], alternative='two-sided')[1]
    print('    share outside the band: device %d / %d vs reference %d / %d, two-sided Fisher p = %.3f' % (out_dev, n, out_ref, n, p_two))
    assert p_two >= 0.05, (out_dev, out_ref, p_two, np.sort(dev)[-30:], np.sort(ref)[-12:])
    assert np.median(dev) <= 1.02 * np.median(ref) and dev.max() <= 1.02 * ref.max(), (np.sort(dev), np.sort(ref))
    assert 0.5 * np.median(sp['ncl32'].sum(1)) <= np.median(ncl) <= 2.0 * np.median(sp['ncl32'].sum(1))
    eng.close()


def test_configs1_fit_spread_against_the_reference_spread():
    """Fit-level parity on the HEADLINE workload as a comparison of distributions (round 6): 4 frames of BASELINE configs[1]
    (seeds 1000 ...; L2 pose prior, yaml stages) from 24 starts each - the bench's start and 23 copies perturbed by 1e-6 -, fitted
    by the reference itself in float32 (tests/golden/configs1_spread.npz, oracle/make_golden_configs1_spread.py) and by the
    device in one batch of 96.  These problems are well conditioned: the reference's 24 fits of a frame end within 3e-5 ...
    7e-4 (relative) of one another.  Per frame: the device's median inside the reference's own [min, max] widened by its width
    (at least 1e-4 of the loss), no device fit worse than the reference's worst by more than that, and a comparable effort."""
    sp = np.load(os.path.join(GOLD, 'configs1_spread.npz'))
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, skin_topk=4, kp_regressor=(d['rows'], d['cols'], d['vals']))
    assert np.array_equal(np.array(syn.model_checksum(model)), sp['model_checksum'])
    nf, ns = sp['final32'].shape
    eng = make_engine(model)
    cams = syn.make_camera_ring(8)
    eng.set_problems(cams, np.repeat(sp['gt_xy'], ns, 0), np.repeat(sp['conf'], ns, 0))
    from mvsmplfitting_amd.engine import stage_weights
    xf, st = eng.fit(sp['x0'].reshape(nf * ns, 118).astype(np.float32), stage_weights(1536.0))
    dev = st['final_loss'].cpu().numpy().astype(np.float64).reshape(nf, ns)
    ncl = st['n_closure'].cpu().numpy().reshape(nf, ns)
    assert np.isfinite(dev).all()
    for f in range(nf):
        ref, rn = sp['final32'][f], sp['ncl32'][f].sum(1)
        # (float32 fits stopped by ftol = 1e-9 scatter at the 1e-5 ... 1e-4 level around an optimum - the reference's 24 ends of frame 1
        # spread over 7e-4, those of frame 0 over 4e-5, the device's over 7e-5: the yard-stick is the reference's own width, at
        # least 1e-4 of the loss)
        width = max(ref.max() - ref.min(), 1e-4 * abs(np.median(ref)))
        print('configs[1] frame %d, %d starts: reference %.4f ... %.4f (median %.4f, closures median %d); device %.4f ... %.4f (median '
              '%.4f, closures median %d)' % (f, ns, ref.min(), ref.max(), np.median(ref), np.median(rn), dev[f].min(), dev[f].max(),
                                             np.median(dev[f]), np.median(ncl[f])))
        assert ref.min() - width <= np.median(dev[f]) <= ref.max() + width, (f, np.sort(dev[f]), np.sort(ref))
        assert dev[f].max() <= ref.max() + width, (f, np.sort(dev[f]), np.sort(ref))
        assert 0.6 * np.median(rn) <= np.median(ncl[f]) <= 1.5 * np.median(rn), (f, ncl[f], rn)
    eng.close()
