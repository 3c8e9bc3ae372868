"""GPU: the asynchronous full-mode fit (mvfit_fit without the SDF term): ONE optimiser kernel runs the staged fit and
publishes the vertex-pass operands of every trial point into a ring; one 6890-vertex LBS pass per closure round runs
concurrently on the other CUs behind a gate kernel that waits for the round's operands.  Checked here:
  * the pass that belongs to closure round r computed the vertices of the trial point of round r (capture hook +
    closure trace), also after the ring wrapped;
  * every closure got its pass: no operand was overwritten before it was read (ring back-pressure: asserted on the first
    and only attempt), no gate timed out;
  * the optimiser is not perturbed: parameters / loss / closure counts equal the objective-vertices-only fit bit for
    bit (the same kernel computes both; the passes only consume what it publishes)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd.engine import stage_weights as eng_stage_weights
from tests.gpu_helpers import make_engine
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu


def _setup(B=5):
    g = dict(np.load(os.path.join(GOLD, 'fit_l2.npz')))
    eng = make_engine(body_model(0, 4))
    cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
    rng = np.random.default_rng(3)
    gt = np.repeat(g['gt_xy'][:1], B, 0) + rng.normal(0, 3.0, (B,) + g['gt_xy'].shape[1:]).astype(np.float32)
    conf = np.repeat(g['conf'][:1], B, 0)
    eng.set_problems(cams, gt, conf)
    x0 = np.zeros((B, 118), np.float32)
    x0[:, 85] = 1.0
    x0[:, :86] += rng.normal(0, 0.02, (B, 86)).astype(np.float32)
    return eng, x0


def _fit_in_step(eng, x0, stages):
    """One fit, no retries: the ring has back-pressure (a problem reuses a slot only after the pass of the round that
    filled it has run), so no pass can lose its operands however the host thread that queues the passes is scheduled."""
    return eng.fit(x0, stages)


@pytest.mark.parametrize('round_index', [0, 9, 131, 200])
def test_pass_of_round_r_computes_the_trial_point_of_round_r(round_index):
    eng, x0 = _setup()
    stages = eng_stage_weights(1536.0, flags=0)
    tr = eng.fit_trace(256)
    cap = eng.capture_pass(round_index)
    xf, st = _fit_in_step(eng, x0, stages)
    eng.capture_pass(None)
    tr = tr.cpu().numpy()
    cap = cap.cpu().numpy()
    ncl = st['n_closure'].cpu().numpy()
    assert st['passes']['missed'] == 0 and st['passes']['timed_out'] == 0
    assert st['passes']['run'] >= ncl.max()                      # one chunk pass per closure round (B <= 32: one chunk)
    assert ncl.max() > 210, ncl                                  # the ring (128 slots) wrapped before round 131 / 200
    eng.fit_trace(0)
    x_r = tr[:, round_index, :118].copy()
    have = round_index < ncl                                      # problems that evaluated a closure in this round
    assert have.any()
    x_r[~have] = x0[~have]
    verts, _ = eng.vertices(x_r)
    verts = verts.cpu().numpy()
    for b in np.flatnonzero(have):
        # same pass kernel, operands from the fit kernel's single-wave chain vs the stand-alone pointer-jumping chain:
        # equal to rounding
        assert np.abs(cap[b] - verts[b]).max() < 2e-6, (b, np.abs(cap[b] - verts[b]).max())
    eng.close()


def test_async_fit_equals_objective_vertices_only_fit():
    eng, x0 = _setup(B=33)
    full = eng_stage_weights(1536.0, flags=0)
    sparse = eng_stage_weights(1536.0, flags=_lib.F_SPARSE_VERTS)
    xa, sa = _fit_in_step(eng, x0, full)
    xs, ss = eng.fit(x0, sparse)
    assert sa['passes']['run'] > 0 and ss['passes']['run'] == 0
    assert sa['passes']['missed'] == 0 and sa['passes']['timed_out'] == 0
    assert np.array_equal(xa.cpu().numpy(), xs.cpu().numpy())
    assert np.array_equal(sa['final_loss'].cpu().numpy(), ss['final_loss'].cpu().numpy())
    assert np.array_equal(sa['n_closure'].cpu().numpy(), ss['n_closure'].cpu().numpy())
    eng.close()


def test_chained_round_mode_still_available():
    """mvfit_options::round_mode = 1: vertex pass -> step kernel per round (the structure the SDF term needs)."""
    eng, x0 = _setup(B=4)
    eng.set_options(round_mode=1)
    xf, st = eng.fit(x0, eng_stage_weights(1536.0, flags=0))
    assert st['passes'] == dict(run=0, skipped=0, missed=0, timed_out=0)
    assert np.all(np.isfinite(st['final_loss'].cpu().numpy()))
    eng.close()


def _capture(eng, x0, stages, round_index, resident):
    eng.set_options(resident_pass=-1 if resident is None else resident)      # -1: automatic
    cap = eng.capture_pass(round_index)
    xf, st = eng.fit(x0, stages)
    eng.capture_pass(None)
    assert st['passes']['missed'] == 0 and st['passes']['timed_out'] == 0, st['passes']
    return cap.cpu().numpy(), xf.cpu().numpy(), st['n_closure'].cpu().numpy(), eng.pass_profile()


@pytest.mark.parametrize('B,resident,tpw,round_index', [(5, None, 1, 9), (5, None, 1, 140), (5, 2, 3, 9), (33, None, 1, 140),
                                                        (70, None, 3, 9), (70, None, 3, 140), (128, None, 3, 31),
                                                        (5, 3, 3, 9), (33, 3, 3, 140), (97, 3, 3, 140)])
def test_resident_pass_is_bit_identical_to_the_per_round_launches(B, resident, tpw, round_index):      # tpw: the expected form
    """The resident pass (one launch per fit, the tiles' basis stationary in registers, rounds served from the ring) writes
    the SAME BITS as the gate + pass launches per closure round: one tile per workgroup (form 1) and two tiles with contraction
    waves + worker waves (form 3, the automatic choice beside more than 36 optimiser workgroups; an explicit 2 - the dropped
    two-tile form of the one-tile kernel - maps to it), one chunk / ragged chunks / four chunks, before and after the ring
    wrapped; and it does not perturb the optimiser either."""
    eng, x0 = _setup(B=B)
    stages = eng_stage_weights(1536.0, flags=0)
    cap_l, x_l, ncl_l, prof_l = _capture(eng, x0, stages, round_index, 0)
    cap_r, x_r, ncl_r, prof_r = _capture(eng, x0, stages, round_index, resident)
    assert prof_l['form'] == 0 and prof_r['form'] == tpw, (prof_l, prof_r)
    assert np.array_equal(x_l, x_r) and np.array_equal(ncl_l, ncl_r)
    have = round_index < ncl_r
    assert have.any()
    assert np.isfinite(cap_r[have]).all()
    assert np.array_equal(cap_l[have], cap_r[have]), np.abs(cap_l[have] - cap_r[have]).max()
    eng.close()


def test_resident_pass_profile_and_alone_timing():
    """mvfit_profile with the resident pass: every round of the fit is stamped inside the kernel (service span, workgroup
    busy time), and the pass can be timed alone over the ring the fit left behind."""
    eng, x0 = _setup(B=32)
    stages = eng_stage_weights(1536.0, flags=0)
    eng.profile(True)
    xf, st = eng.fit(x0, stages)
    pr = eng.profile_read()
    pp = eng.pass_profile()
    eng.profile(False)
    ncl = int(st['n_closure'].max().item())
    assert pp['tiles_per_workgroup'] == 1 and pp['workgroups'] == 216
    assert pp['rounds_stamped'] == min(ncl, 1024) == pr['vertex_pass_launches'], (pp, ncl, pr)
    assert 0.0 < pp['workgroup_busy_ms'] <= pp['slowest_workgroup_ms'] <= pp['round_span_ms'] < 0.05, pp
    assert pr['vertex_pass_ms'] == pp['round_span_ms']
    alone = eng.profile_resident_pass_ms(100)
    assert 0.0 < alone < 0.05, alone
    # the ring is usable again afterwards: same fit, same result
    xf2, st2 = eng.fit(x0, stages)
    assert np.array_equal(xf.cpu().numpy(), xf2.cpu().numpy())
    assert st2['passes']['missed'] == 0 and st2['passes']['timed_out'] == 0
    eng.close()


def test_compact_and_two_loop_direction_forms_end_on_the_same_fits():
    """The single-launch fit takes the L-BFGS direction in compact (Byrd-Nocedal-Schnabel) form with an explicitly maintained
    float32 R^-1, the chained step kernel in two-loop form over its Gram matrices: two associations of the same direction
    (lbfgs_ls.py:336-358).  Whole staged fits of the same problems must end on the same optima to the precision two float32
    roundings of a 300-closure trajectory allow (a poorly conditioned R^-1 would show as a worse fit or an early stop)."""
    eng, x0 = _setup(B=16)
    stages = eng_stage_weights(1536.0, flags=0)
    xa, sa = eng.fit(x0, stages)
    eng.set_options(round_mode=1)
    xc, sc = eng.fit(x0, stages)
    eng.set_options(round_mode=0)
    fa, fc = sa['final_loss'].cpu().numpy().astype(np.float64), sc['final_loss'].cpu().numpy().astype(np.float64)
    na, nc = sa['n_closure'].cpu().numpy(), sc['n_closure'].cpu().numpy()
    rel = np.abs(fa - fc) / np.abs(fc)
    print('compact vs two-loop: final loss rel diff median %.1e max %.1e; closures %d vs %d' % (np.median(rel), rel.max(), na.sum(), nc.sum()))
    assert np.isfinite(fa).all() and np.isfinite(fc).all()
    assert np.median(rel) <= 2e-3 and rel.max() <= 5e-2, rel
    assert 0.7 * nc.sum() <= na.sum() <= 1.3 * nc.sum(), (na, nc)
    eng.close()


@pytest.mark.parametrize('resident', [-1, 0])
def test_round_cap_ends_the_fit_with_an_error_and_leaves_the_engine_usable(resident):
    """max_rounds smaller than the fit needs: mvfit_fit returns MVFIT_E_STATE at once (the optimiser kernel stops at the cap,
    every problem tells the passes that nothing more comes, the resident pass / the queued per-round passes end) - no hang,
    no time-out - and the next uncapped fit is the fit it always was."""
    from mvsmplfitting_amd.engine import MvFitError
    eng, x0 = _setup(B=6)
    eng.set_options(resident_pass=resident)
    stages = eng_stage_weights(1536.0, flags=0)
    x_ref, s_ref = eng.fit(x0, stages)
    import time
    t0 = time.time()
    with pytest.raises(MvFitError, match='round cap'):
        eng.fit(x0, stages, max_rounds=48)
    assert time.time() - t0 < 5.0
    x2, s2 = eng.fit(x0, stages)
    assert np.array_equal(x_ref.cpu().numpy(), x2.cpu().numpy())
    assert s2['passes']['missed'] == 0 and s2['passes']['timed_out'] == 0
    eng.close()
