"""GPU: the float32 PRODUCTION fit (mvfit_fit: HIP closure + device L-BFGS state machine) followed closure for closure
against the reference's own float32 fit on the real objective (traces recorded by oracle/make_golden.py /
make_golden_demo.py: the (x, loss) of every closure call of create_fitting_closure + LBFGSLs + run_fitting).

Both runs are float32 programs with different summation orders, so they agree to rounding at the first closure and
drift apart as the line search amplifies last-bit differences (the reference's own float32 and float64 runs - both in
the goldens - drift the same way: 1e-7 at the start, 1e-5..1e-4 after ~30 closures, a different branch somewhere
between closure 35 and 50; on the demo's ill-conditioned start after ~8).  The assertion: over the first outer step
the device trace stays within TOL(k) of the reference float32 trace, TOL growing geometrically from 2e-5 to 3e-3
(measured: 1e-8 ... 3e-5 on x, up to 1.4e-5 relative on the loss of a far line-search trial; tests/report_traj.py) - a
wrong branch in the state machine (bracket / zoom / cubic step / history update) shows up as an O(1e-1) jump at the
closure where it happens."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import stage_weights as eng_stage_weights
from tests.gpu_helpers import from118, make_engine, to118
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu
N_STEP = 35                 # closures of the first outer step (max_eval = 37)


def tol(k, n=N_STEP, lo=2e-5, hi=3e-3):
    return lo * (hi / lo) ** (min(k, n - 1) / (n - 1))


def _compare(trace_dev, ref32, ref64, use_vp, n):
    """max over k < n of err(k) / TOL(k) for x (absolute, parameters are O(0.1..1)) and the loss (relative)."""
    worst = 0.0
    rows = []
    for k in range(n):
        x = from118(trace_dev[k, :118], use_vp)
        ex = np.abs(x - ref32[k, :-1]).max()
        el = abs(trace_dev[k, 118] - ref32[k, -1]) / abs(ref32[k, -1])
        ex_ref = np.abs(ref64[k, :-1] - ref32[k, :-1]).max()
        rows.append((k, ex, el, ex_ref))
        worst = max(worst, ex / tol(k), el / tol(k))
    return worst, rows


@pytest.mark.parametrize('name,use_vp', [('l2', False), ('vposer', True)])
@pytest.mark.parametrize('sparse', [False, True])
def test_fp32_fit_follows_reference_fp32_trajectory(name, use_vp, sparse):
    g = dict(np.load(os.path.join(GOLD, 'fit_%s.npz' % name)))
    eng = make_engine(body_model(), syn.make_vposer_decoder() if use_vp else None)
    cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
    B = g['x0'].shape[0]
    eng.set_problems(cams, g['gt_xy'], g['conf'])
    x0 = np.stack([to118(g['x0'][b], use_vp) for b in range(B)]).astype(np.float32)
    flags = (_lib.F_VPOSER if use_vp else 0) | (_lib.F_SPARSE_VERTS if sparse else 0)
    stages = eng_stage_weights(1536.0, flags=flags)
    tr = eng.fit_trace(120)
    xf, st = eng.fit(x0, stages)
    tr = tr.cpu().numpy().astype(np.float64)
    eng.fit_trace(0)
    ncl = st['n_closure'].cpu().numpy()
    final = st['final_loss'].cpu().numpy().astype(np.float64)
    for b in range(B):
        assert np.isfinite(tr[b, :min(120, ncl[b])]).all() and np.isnan(tr[b, min(120, ncl[b]):]).all()
        n = min(N_STEP, int(g['ncl32'][b][0]), int(g['ncl'][b][0]))          # inside the first stage of both reference runs
        worst, rows = _compare(tr[b], g['trace32'][b], g['trace64'][b], use_vp, n)
        assert worst <= 1.0, (name, b, worst, [r for r in rows if r[1] > tol(r[0]) or r[2] > tol(r[0])][:5])
        # end quality next to the reference's float32 and float64 fits
        ref_hi = max(float(g['final'][b]), float(g['final32'][b]))
        assert final[b] <= 1.05 * ref_hi, (name, b, final[b], ref_hi)
    eng.close()


def test_demo_fit_follows_reference_fp32_trajectory():
    """configs[0] (real cameras / keypoints / VPoser checkpoint): the start is ill-conditioned (scale 2, translation
    10) and the reference's own float32 and float64 runs part ways at closure 7 (5e-2 apart there, 1e-6 before); the
    first 7 closures are compared."""
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    eng = make_engine(body_model(), vpw)
    cams = tuple(g[k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    eng.set_problems(cams, g['gt_xy'][None], g['conf'][None])
    stages = [dict(data_weight=float(w[0]), body_pose_weight=float(w[1]), shape_weight=float(w[2]),
                   bending_prior_weight=float(w[3]), rho=float(w[4]), flags=_lib.F_VPOSER) for w in g['stage_w']]
    tr = eng.fit_trace(16)
    eng.fit(to118(g['x0'], True)[None].astype(np.float32), stages)
    tr = tr.cpu().numpy().astype(np.float64)[0]
    eng.close()
    worst, rows = _compare(tr, g['fit_trace32'], g['fit_trace64'], True, 7)
    # x here is O(10) (translation): scale the absolute tolerance accordingly
    worst = max(max(r[1] / (10 * tol(r[0], 7)), r[2] / tol(r[0], 7)) for r in rows)
    assert worst <= 1.0, [(r[0], float('%.2g' % r[1]), float('%.2g' % r[2]), float('%.2g' % r[3])) for r in rows]


@pytest.mark.parametrize('service', [1, 0])
@pytest.mark.parametrize('name', ['l2_s3_v6', 'l2_top4_v8'])
def test_fit_whose_every_round_carries_the_sdf_term_follows_the_reference(name, service):
    """Two stages that BOTH carry the interpenetration term (yaml stage-3 / stage-4 weights, coll_loss_weights 1000 / 4500),
    from a body with a vertex in the first triangle's shadow: every closure of the device fit is a round that needs the term -
    service = 1 (round 6, the default): the single-launch optimiser kernel asks for it every round (gate -> vertex pass -> term
    kernels on the pass stream, the pull-back answers through memory); service = 0: the CHAINED structure (vertex pass -> term
    kernels -> step kernel launch per round), kept as the checker.  Reference: its own float32 fit with its own
    SMPLifyLoss(interpenetration=True), every closure call recorded (tests/golden/fit_sdf.npz, oracle/make_golden_sdf_fit.py;
    float32 is the only precision the reference's term runs in).  Four checks:
      (a) the device closure at each of the reference's 120 recorded trial points returns the reference's loss (2e-5; 3e-4
          where the term is non-zero: pen = (1000 S)^2 amplifies the last bits of the vertex positions - measured <= 4e-5);
      (b) every round of the chained fit returned what the closure call returns at that round's trial point (1e-6) - the
          chained structure evaluates the function (a) pins, at the point the optimiser asked for;
      (c) the device trajectory follows the reference's (the tolerance schedule of the tests above) - until the term, which
          is only piecewise continuous (a vertex entering or leaving the triangle's shadow moves S by tens of per cent; trial
          points 2e-6 apart give S values 2e-4 apart), has differed between the two runs' trial points by more than 5e-5; it
          may leave the reference's trajectory only behind such a point (measured: closure 3 and closure 25);
      (d) the fit ends where the reference's ends (final loss <= 1.05 x; measured 449.2725 vs 449.2715 and 553.49 vs 553.33,
          closure counts 376 vs 351 and 171 vs 218)."""
    from tests.helpers import load_case
    t = dict(np.load(os.path.join(GOLD, 'fit_sdf.npz')))
    cfg, g, model, vpw, gmm, wts, _ = load_case(name)
    assert abs(syn.model_checksum(model) - float(t[name + '/model_checksum'])) < 1e-6 * float(t[name + '/model_checksum'])
    eng = make_engine(model, None, None, sdf_service=service)
    cams = tuple(t[name + '/' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    eng.set_problems(cams, t[name + '/gt_xy'][None], t[name + '/conf'][None])
    eng.set_sdf(model['faces'], num_faces=1, grid_size=128)
    x0 = to118(t[name + '/x0'], False)[None].astype(np.float32)
    stages = eng_stage_weights(1536.0, coll_w=[0.0, 0.0, float(t['coll_w'][0]), float(t['coll_w'][1])])[2:4]
    ref = t[name + '/trace32']
    nref = min(ref.shape[0], int(t[name + '/ncl32'][0]))              # recorded closures of the first stage

    def closure_at(x118):
        out = eng.closure(x118[None].astype(np.float32), stages[0], want_grad=False)
        _, S = eng.sdf_term_read()
        return float(out['loss'][0]), float(S[0])

    # (a)
    S_ref = np.zeros(nref)
    for k in range(nref):
        L, S_ref[k] = closure_at(to118(ref[k, :-1], False))
        assert abs(L - ref[k, -1]) <= (3e-4 if S_ref[k] > 0 else 2e-5) * abs(ref[k, -1]), (name, k, L, ref[k, -1], S_ref[k])
    assert (S_ref > 0).sum() >= 3, 'the case does not exercise the term'
    tr = eng.fit_trace(120)
    xf, st = eng.fit(x0, stages)
    tr = tr.cpu().numpy().astype(np.float64)[0]
    eng.fit_trace(0)
    ndev = min(120, int(st['n_closure'][0]))
    # (b)
    S_dev = np.zeros(ndev)
    for k in range(ndev):
        L, S_dev[k] = closure_at(tr[k, :118])
        # service rounds: the vertices of the trial point come from the optimiser kernel's own pose / chain code (the objective's 69
        # directly, all 6890 through the operands it publishes), the closure call's and the chained rounds' from the stand-alone
        # prep code - two roundings, 2e-6 apart on the vertices (tests/test_gpu_async.py), which the term amplifies (docstring (c):
        # trial points 2e-6 apart give S values 2e-4 apart): 1e-5 of the loss (north_star) + 4e-4 of the penalty (w S)^2
        pen = (float(stages[0]['coll_loss_weight']) * S_dev[k]) ** 2
        assert abs(L - tr[k, 118]) <= ((1e-5 * abs(L) + 4e-4 * pen) if service else 1e-6 * abs(L)), (name, service, k, L, tr[k, 118], pen)
    # (c)
    n = min(N_STEP, nref, ndev)
    ex = np.array([np.abs(from118(tr[k, :118], False) - ref[k, :-1]).max() for k in range(n)])
    off = [k for k in range(n) if ex[k] > tol(k)]
    if off:
        jumps = [k for k in range(off[0]) if abs(S_dev[k] - S_ref[k]) > 5e-5 * max(S_dev[k], S_ref[k], 1e-6)]
        assert off[0] >= 2 and jumps, (name, off[0], ex[:off[0] + 1], S_dev[:off[0] + 1], S_ref[:off[0] + 1])
    # (d)
    final = float(st['final_loss'][0])
    assert final <= 1.05 * float(t[name + '/final32']), (name, final, float(t[name + '/final32']))
    eng.close()


@pytest.mark.parametrize('service', [1, 0])
@pytest.mark.parametrize('key', ['vp_s0_v8', 'vp_s0_v8/yaml4'])
def test_vposer_fits_with_the_sdf_term_follow_the_reference(key, service):
    """Round 6: the VPoser branch (cfg_files/fit_smpl.yaml:35-37) with the interpenetration term, recorded from the reference's own
    float32 fit like the cases above (oracle/make_golden_sdf_fit.py -> tests/golden/fit_sdf.npz): 'vp_s0_v8' = the two stages that
    carry the term (yaml stage-3 / stage-4 weights) from a body with a vertex in the first triangle's shadow; 'vp_s0_v8/yaml4' = ALL
    FOUR yaml stages (coll_loss_weights 0, 0, 1000, 4500) from the same start - on the device the lead stages (single launch,
    resident pass, decoder helpers), the hand-over at the stage boundary and the rounds that ask for the term (service = 1) or the
    chained rounds (service = 0) in one fit.
      (a) the device closure returns the reference's loss at its recorded trial points (every stage's own weights);
      (b) the device trajectory follows the reference's over the first outer step (tolerance schedule of the tests above) or
          leaves it only behind a point where the piecewise-continuous term differed;
      (c) the fit ends where the reference's ends."""
    from tests.helpers import load_case
    t = dict(np.load(os.path.join(GOLD, 'fit_sdf.npz')))
    name = key.split('/')[0]
    cfg, g, model, vpw, gmm, wts, _ = load_case(name)
    assert abs(syn.model_checksum(model) - float(t[key + '/model_checksum'])) < 1e-6 * float(t[key + '/model_checksum'])
    eng = make_engine(model, vpw, None, sdf_service=service)
    cams = tuple(t[key + '/' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    eng.set_problems(cams, t[key + '/gt_xy'][None], t[key + '/conf'][None])
    eng.set_sdf(model['faces'], num_faces=1, grid_size=128)
    x0 = to118(t[key + '/x0'], True)[None].astype(np.float32)
    sidx = [int(i) for i in t[key + '/stage_index']]
    allst = eng_stage_weights(1536.0, flags=_lib.F_VPOSER, coll_w=[0.0, 0.0, float(t['coll_w'][0]), float(t['coll_w'][1])])
    stages = [allst[i] for i in sidx]
    ref, ncl_ref = t[key + '/trace32'], [int(n) for n in t[key + '/ncl32']]
    bounds = np.cumsum(ncl_ref)
    # (a)
    n_term = 0
    for k in range(ref.shape[0]):
        st_k = stages[min(int(np.searchsorted(bounds, k, side='right')), len(stages) - 1)]
        out = eng.closure(to118(ref[k, :-1], True)[None].astype(np.float32), st_k, want_grad=False)
        L = float(out['loss'][0])
        S = eng.sdf_term_read()[1] if st_k['coll_loss_weight'] > 0 else [0.0]
        hit = st_k['coll_loss_weight'] > 0 and float(S[0]) > 0
        n_term += int(hit)
        assert abs(L - ref[k, -1]) <= (3e-4 if hit else 2e-5) * abs(ref[k, -1]), (key, k, L, ref[k, -1], float(S[0]))
    assert n_term >= 3, 'the case does not exercise the term'
    tr = eng.fit_trace(120)
    xf, st = eng.fit(x0, stages)
    tr = tr.cpu().numpy().astype(np.float64)[0]
    eng.fit_trace(0)
    assert st['passes']['missed'] == 0 and st['passes']['timed_out'] == 0
    assert st['decoder']['answers_timed_out'] == 0 and st['decoder']['helpers_gave_up'] == 0
    # (b)
    n = min(N_STEP, ref.shape[0], int(st['n_closure'][0]), ncl_ref[0])
    ex = np.array([np.abs(from118(tr[k, :118], True) - ref[k, :-1]).max() for k in range(n)])
    off = [k for k in range(n) if ex[k] > tol(k)]
    if off:
        jumps = []
        for k in range(off[0] if stages[0]['coll_loss_weight'] > 0 else 0):
            Sd = []
            for xk in (tr[k, :118], to118(ref[k, :-1], True)):
                eng.closure(xk[None].astype(np.float32), stages[0], want_grad=False)
                Sd.append(float(eng.sdf_term_read()[1][0]))
            if abs(Sd[0] - Sd[1]) > 5e-5 * max(Sd[0], Sd[1], 1e-6):
                jumps.append(k)
        print('%s service=%d: leaves the reference trajectory at closure %d (x error %s), term differed at closures %s'
              % (key, service, off[0], ex[:off[0] + 1], jumps))
        # (the term is active from the first closure on and dominates the loss there - 3.2e5 against 1.2e4 at the end -: its float32
        # gradient is known to 2e-3 of its maximum on both sides (tests/test_gpu_sdf_term.py), so already the first step can differ
        # by more than the no-term schedule's 2e-5)
        if stages[0]['coll_loss_weight'] > 0:
            assert off[0] >= 1 and (jumps or ex[off[0]] <= 50 * tol(off[0])), (key, off[0], ex[:off[0] + 1])
        else:
            # 'yaml4': the first outer step is a stage WITHOUT the term, i.e. the code path of test_fp32_fit_follows_reference_fp32_
            # trajectory's VPoser case, from a much rougher start (loss 8.5e6, a pose drawn to reach the triangle's shadow): two
            # float32 programs drift apart faster there - measured 8.7e-5 at closure 7 against the schedule's 5.6e-5 - but a wrong
            # branch still shows as an O(1e-1) jump: five times the schedule
            assert max(ex[k] / tol(k) for k in range(n)) <= 5.0, (key, off[0], ex[:n])
    # (c) these fits have several optima (3.6 k / 4.6 k / 6.7 k / 12.2 k / 13.8 k) and the last bits of a float32 program decide
    #     which one it ends on: the reference's OWN float32 fits from 24 starts perturbed by 1e-6 (oracle/make_golden_sdf_fit.py
    #     spread -> tests/golden/fit_sdf_vp_spread.npz) are the yard-stick - the device's fits from the same 24 starts (one batch)
    #     must not be worse than the reference's: median <= 1.05 x its median, worst <= 1.02 x its worst
    final, fref = float(st['final_loss'][0]), float(t[key + '/final32'])
    sp = np.load(os.path.join(GOLD, 'fit_sdf_vp_spread.npz'))
    x0s, fsp = sp[key + '/x0'], sp[key + '/final32']
    nb = x0s.shape[0]
    eng.set_problems(cams, np.repeat(t[key + '/gt_xy'][None], nb, 0), np.repeat(t[key + '/conf'][None], nb, 0))
    eng.set_sdf(model['faces'], num_faces=1, grid_size=128)
    xf, stb = eng.fit(np.stack([to118(x, True) for x in x0s]).astype(np.float32), stages)
    dev = stb['final_loss'].cpu().numpy().astype(np.float64)
    print('%s service=%d: device final %.1f (%d closures), reference float32 %.1f (%d closures); %d starts: device median %.1f max %.1f, '
          'reference median %.1f max %.1f' % (key, service, final, int(st['n_closure'][0]), fref, sum(ncl_ref), nb, np.median(dev), dev.max(),
                                              np.median(fsp), fsp.max()))
    assert np.isfinite(final) and final <= 1.02 * fsp.max(), (key, final, np.sort(fsp))
    assert np.isfinite(dev).all() and np.median(dev) <= 1.05 * np.median(fsp) and dev.max() <= 1.02 * fsp.max(), (key, np.sort(dev), np.sort(fsp))
    assert stb['passes']['missed'] == 0 and stb['passes']['timed_out'] == 0
    eng.close()
