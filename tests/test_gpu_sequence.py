"""GPU: sequence mode for a batch of sequences (mvsmplfitting_amd.sequence; reference is_seq: main.py:76-79,
init_guess.py:137-166, non_linear_solver.py:158-162) against the reference's own chain recorded in
tests/golden/sequence_ref.npz (oracle/make_golden_sequence.py: the reference's load_init + fix_params +
non_linear_solver on synthetic sequences, float32 and float64):
  * every frame fitted from the reference's recorded start vector with the reference's stage list ends where the
    reference ended (final loss <= 1.05 x the worse of its float32 / float64 runs);
  * the wavefront over whole chains takes the reference's cold / warm decisions (incl. the restart after the frame
    whose loss exceeded 5000) and stays inside the same bound;
  * (the carry-over rule itself - start vector and stage weights, exact - is tests/test_sequence_ref.py, CPU);
  * the wavefront equals fitting every chain by hand, frame after frame (scheduling only changes the batching)."""
import numpy as np
import pytest

from mvsmplfitting_amd import sequence as sq
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import stage_weights
from oracle import closure_np as cn
from tests.gpu_helpers import make_engine
from tests.helpers import body_model

pytestmark = pytest.mark.gpu


def _sequences(S=3, T=4, V=6):
    model = body_model(0, 4)
    cams = syn.make_camera_ring(V)
    orc = cn.ClosureOracle(model, np.float64)
    gt = np.zeros((S, T, V, 17, 2), np.float32)
    cf = np.zeros((S, T, V, 17), np.float32)
    for s in range(S):
        fr = syn.make_frames(1, seed0=300 + s)
        base = {k: fr[k][0].astype(np.float64) for k in fr}
        for t in range(T):                                    # a slow motion: the pose drifts from frame to frame
            p = dict(base, use_vposer=False)
            p['body_pose'] = base['body_pose'] + 0.02 * t
            p['transl'] = base['transl'] + np.array([0.01 * t, 0.0, 0.0])
            kp = orc.body(p, want_cache=False)['joints']
            g, c = syn.make_observations(kp[None], cams, seed=10 * s + t)
            gt[s, t], cf[s, t] = g[0], c[0]
    x_init = np.zeros((S, T, 118), np.float32)
    x_init[..., 85] = 1.0
    x_init[..., 13:19] = 1.0                                  # fix_params' body pose start (init_guess.py:199-203)
    return model, cams, gt, cf, x_init


def _ref_chains(names, use_vp):
    from tests.test_sequence_ref import load_chain, x_init_118
    from tests.gpu_helpers import to118
    c32 = [load_chain(n, 'float32') for n in names]
    c64 = [load_chain(n, 'float64') for n in names]
    cams = tuple(c32[0][k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    gt = np.stack([c['gt_xy'] for c in c32]); cf = np.stack([c['conf'] for c in c32])
    xi = np.stack([x_init_118(c, use_vp) for c in c32])
    x0 = np.stack([[to118(x, use_vp) for x in c['x0']] for c in c32]).astype(np.float32)
    bound = 1.05 * np.maximum(np.stack([c['loss'] for c in c32]), np.stack([c['loss'] for c in c64]))
    cold = np.stack([c['seq_start'] for c in c32]).astype(bool)
    return c32, cams, gt, cf, xi, x0, bound, cold


@pytest.mark.parametrize('use_vp', [False, True])
def test_frames_from_the_references_start_vectors_end_where_the_reference_ended(use_vp):
    from mvsmplfitting_amd import _lib
    names = ['vp_a'] if use_vp else ['l2_a', 'l2_b']
    c32, cams, gt, cf, xi, x0, bound, cold = _ref_chains(names, use_vp)
    model = body_model(0, 4)
    assert abs(syn.model_checksum(model) - c32[0]['model_checksum']) < 1e-6 * c32[0]['model_checksum']
    vpw = syn.make_vposer_decoder(seed=3, gain=1.0, identity_bias=True) if use_vp else None
    eng = make_engine(model, vpw)
    full = stage_weights(1536.0, flags=_lib.F_VPOSER if use_vp else 0)
    warm = sq.sequence_stages(full)
    for sel, stg in ((cold, full), (~cold, warm)):
        s_idx, t_idx = np.nonzero(sel)
        eng.set_problems(cams, gt[s_idx, t_idx], cf[s_idx, t_idx])
        xf, st = eng.fit(x0[s_idx, t_idx], stg)
        fl = st['final_loss'].cpu().numpy()
        assert np.all(np.isfinite(fl))
        assert np.all(fl <= bound[s_idx, t_idx]), (fl, bound[s_idx, t_idx])
    eng.close()


def test_wavefront_takes_the_references_decisions():
    c32, cams, gt, cf, xi, x0, bound, cold = _ref_chains(['l2_a', 'l2_b'], False)
    eng = make_engine(body_model(0, 4))
    xs, st = sq.fit_sequences(eng, cams, gt, cf, xi, stage_weights(1536.0, flags=0))
    fl = st['final_loss'].cpu().numpy(); ncl = st['n_closure'].cpu().numpy()
    assert np.array_equal(st['restarted'], cold), (st['restarted'], cold)       # incl. the restart behind the 77 k frame
    assert fl[1, 1] > sq.RESTART_LOSS
    assert np.all(fl <= bound), (fl, bound)
    xs = xs.cpu().numpy()
    # frame 0 started from the reference's start vector; a warm frame from its own predecessor by the carry-over rule
    assert np.array_equal(xi[:, 0], x0[:, 0])
    assert ncl[~cold].mean() < 0.9 * ncl[cold].mean()
    eng.close()


def test_wavefront_equals_the_chains_by_hand():
    model, cams, gt, cf, x_init = _sequences()
    S, T = gt.shape[:2]
    stages = stage_weights(1536.0, flags=0)
    eng = make_engine(model)
    xs, st = sq.fit_sequences(eng, cams, gt, cf, x_init, stages)
    xs = xs.cpu().numpy(); fl = st['final_loss'].cpu().numpy(); ncl = st['n_closure'].cpu().numpy()
    assert st['restarted'][:, 0].all() and not st['restarted'][:, 1:].any()
    assert ncl[:, 1:].mean() < 0.9 * ncl[:, 0].mean()         # warm frames run two of the four stages (the last one dominates)
    warm = sq.sequence_stages(stages)
    assert len(warm) == 2 and abs(warm[0]['body_pose_weight'] - 57.4 * 0.15) < 1e-6 and warm[1] == stages[3]
    for s in range(S):                                        # one sequence at a time, frame after frame
        prev = None
        for t in range(T):
            eng.set_problems(cams, gt[s:s + 1, t], cf[s:s + 1, t])
            if t == 0:
                x0, stg = x_init[s:s + 1, t], stages
            else:
                x0 = prev.copy(); x0[:, 13:82] = x_init[s, t, 13:82]; stg = warm
            xf, o = eng.fit(x0, stg)
            prev = xf.cpu().numpy()
            assert np.array_equal(prev[0], xs[s, t]) and o['final_loss'].cpu().numpy()[0] == fl[s, t]
    eng.close()


def test_restart_rule(monkeypatch):
    model, cams, gt, cf, x_init = _sequences(S=2, T=2)
    eng = make_engine(model)
    monkeypatch.setattr(sq, 'RESTART_LOSS', 1.0)              # every previous loss is "too large": init_guess.py:141-145
    xs, st = sq.fit_sequences(eng, cams, gt, cf, x_init, stage_weights(1536.0, flags=0))
    assert st['restarted'].all()
    ncl = st['n_closure'].cpu().numpy()
    assert np.all(ncl[:, 1] > 0.5 * ncl[:, 0])                # the full four-stage schedule again
    eng.close()
