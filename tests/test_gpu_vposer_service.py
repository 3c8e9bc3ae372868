"""GPU: the VPoser decoder helpers of the single-launch fits (csrc/vposer_service.h): the decoder's three layers
(code/model/VPoser.py:218-232) evaluated on helper workgroups with register-resident weights instead of in every
problem's own workgroup.  The parity pin of that arithmetic is tests/test_gpu_trajectory.py (the helpers are on by
default: the production fit follows the reference's own float32 trajectory closure for closure); here: the helpers
really run, nothing times out, the fit agrees with the in-workgroup decoder as closely as two summation orders of the
same float32 products can (first closures to rounding, then the usual line-search amplification), and a problem's
result does not depend on the batch it is fitted in (set / slot / sub-batch)."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights

pytestmark = pytest.mark.gpu


def _problems(eng, B, views=8, seed=5):
    cams = syn.make_camera_ring(views)
    fr = syn.make_frames(B, seed0=seed)
    x = np.zeros((B, 118), np.float32)
    for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items():
        x[:, a:b] = fr[k]
    eng.set_problems(cams, np.zeros((B, views, 17, 2), np.float32), np.ones((B, views, 17), np.float32))
    _, joints = eng.vertices(x)
    gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=seed + 7)
    eng.set_problems(cams, gt, conf)
    x0 = np.zeros((B, 118), np.float32)
    x0[:, 85] = 1.0
    return x0, cams, gt, conf


def _fit(eng, x0, flags, helpers, trace=0):
    eng.set_options(vposer_helpers=1 if helpers else 0)
    tr = eng.fit_trace(trace) if trace else None
    xf, st = eng.fit(x0, stage_weights(1536.0, flags=flags))
    ds = eng.decoder_stats()
    out = dict(x=xf.cpu().numpy(), final=st['final_loss'].cpu().numpy(), ncl=st['n_closure'].cpu().numpy(), stats=ds,
               passes=st['passes'], trace=None if tr is None else tr.cpu().numpy().astype(np.float64))
    if trace:
        eng.fit_trace(0)
    return out


@pytest.mark.parametrize('sparse', [True, False])
def test_helpers_run_and_agree_with_the_in_workgroup_decoder(sparse):
    eng = MvFit(syn.make_body_model(0, skin_topk=4), vposer=syn.make_vposer_decoder())
    B = 11                                                  # 8 sets, three of them with two problems
    x0 = _problems(eng, B)[0]
    flags = _lib.F_VPOSER | (_lib.F_SPARSE_VERTS if sparse else 0)
    a = _fit(eng, x0, flags, True, trace=40)
    b = _fit(eng, x0, flags, False, trace=40)
    assert a['stats'] == dict(launches=1, answers_timed_out=0, helpers_gave_up=0), a['stats']
    assert b['stats']['launches'] == 0
    assert a['passes']['missed'] == 0 and a['passes']['timed_out'] == 0
    for k in range(12):                 # rounding at the first closures, then the line search amplifies (cf. test_gpu_trajectory.tol)
        rel = 3e-6 if k == 0 else (2e-5 if k < 3 else 5e-3)
        la, lb = a['trace'][:, k, 118], b['trace'][:, k, 118]
        assert np.all(np.abs(la - lb) <= rel * np.abs(lb)), (k, la, lb)
        assert np.abs(a['trace'][:, k, :118] - b['trace'][:, k, :118]).max() <= (0.0 if k == 0 else 50 * rel), k
    assert np.all(np.isfinite(a['final'])) and np.all(a['final'] <= 1.5 * b['final'] + 1.0), (a['final'], b['final'])
    eng.close()


def test_a_problems_fit_does_not_depend_on_the_batch_around_it():
    """33 problems (sets of 4-5 problems, slots 0..4) against the same problems fitted 3 at a time (sets of one): same
    helper arithmetic whatever the set, the slot or the number of problems a helper serves - bit for bit."""
    eng = MvFit(syn.make_body_model(0, skin_topk=4), vposer=syn.make_vposer_decoder())
    B = 33
    flags = _lib.F_VPOSER | _lib.F_SPARSE_VERTS
    x0, cams, gt, conf = _problems(eng, B)
    big = _fit(eng, x0, flags, True)
    assert big['stats']['answers_timed_out'] == 0 and big['stats']['helpers_gave_up'] == 0
    for lo in (0, 15, 30):
        eng.set_problems(cams, gt[lo:lo + 3], conf[lo:lo + 3])
        small = _fit(eng, x0[lo:lo + 3], flags, True)
        assert np.array_equal(small['x'], big['x'][lo:lo + 3]), lo
        assert np.array_equal(small['final'], big['final'][lo:lo + 3])
        assert np.array_equal(small['ncl'], big['ncl'][lo:lo + 3])
    eng.close()


@pytest.mark.parametrize('B,sparse,launches', [(150, True, 1), (96, False, 1), (170, True, 2), (100, False, 2)])
def test_helpers_serving_many_problems_each(B, sparse, launches):
    """One launch with 8 sets: 150 problems objective-only = 19 per helper (every polling wave watches three slots), 96
    asynchronous = 12 per helper; 170 / 100 problems = two sub-batches (all workgroups of a launch must be resident).
    No time-out, and a sample of the problems equals the same problems fitted alone."""
    eng = MvFit(syn.make_body_model(0, skin_topk=4), vposer=syn.make_vposer_decoder())
    flags = _lib.F_VPOSER | (_lib.F_SPARSE_VERTS if sparse else 0)
    x0, cams, gt, conf = _problems(eng, B, seed=11)
    big = _fit(eng, x0, flags, True)
    assert big['stats'] == dict(launches=launches, answers_timed_out=0, helpers_gave_up=0), big['stats']
    assert np.all(np.isfinite(big['final']))
    assert big['passes']['missed'] == 0 and big['passes']['timed_out'] == 0
    for lo in (0, 70, B - 2):
        eng.set_problems(cams, gt[lo:lo + 2], conf[lo:lo + 2])
        small = _fit(eng, x0[lo:lo + 2], flags, True)
        assert np.array_equal(small['x'], big['x'][lo:lo + 2]), lo
        assert np.array_equal(small['final'], big['final'][lo:lo + 2])
    eng.close()


def test_a_fit_whose_helpers_never_answer_falls_back_to_the_local_decoder():
    """Fault injection (the -DMVFIT_DEBUG_HOOKS build, MVFIT_VP_FAULT=1: the helper workgroups leave at once): every
    problem's first request times out (50 ms), the problem decodes in its own workgroup from then on - the fit completes,
    the counters say what happened, and the result is the helpers-off fit bit for bit.  The released library has no such
    hook (and reads no environment variable): the test loads the hooks build."""
    hooks = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mvsmplfitting_amd', 'libmvfit_hooks.so')
    if not os.path.isfile(hooks):
        pytest.skip('libmvfit_hooks.so not built (make -C mvsmplfitting_amd/csrc hooks)')
    eng = MvFit(syn.make_body_model(0, skin_topk=4), vposer=syn.make_vposer_decoder(), library=hooks)
    B = 5
    x0 = _problems(eng, B)[0]
    flags = _lib.F_VPOSER | _lib.F_SPARSE_VERTS
    off = _fit(eng, x0, flags, False)
    os.environ['MVFIT_VP_FAULT'] = '1'
    try:
        bad = _fit(eng, x0, flags, True)
    finally:
        del os.environ['MVFIT_VP_FAULT']
    assert bad['stats']['launches'] == 1 and bad['stats']['answers_timed_out'] == B and bad['stats']['helpers_gave_up'] == 5 * 8, bad['stats']
    assert np.array_equal(bad['x'], off['x']) and np.array_equal(bad['final'], off['final']) and np.array_equal(bad['ncl'], off['ncl'])
    eng.close()


def test_vposer_fit_of_32_problems_takes_the_resident_pass_and_writes_the_same_bits():
    """The reference's default mode (cfg_files/fit_smpl.yaml:35-37, use_vposer) at <= 32 problems: 16 helper sets would leave no
    CUs for the resident vertex pass (32 + 128 + 108 workgroups), so the automatic choice takes 8 sets and the role-split
    resident pass (round 6; before, this mode ran its passes as per-round launches).  Against resident_pass = 0 (16 sets, gate +
    pass launch per round): the same fit bit for bit - a problem's result does not depend on the number of sets - and the SAME
    vertices from the pass of a round before and of a round after the ring wrapped."""
    eng = MvFit(syn.make_body_model(0, skin_topk=4), vposer=syn.make_vposer_decoder())
    B = 32
    x0 = _problems(eng, B)[0]
    stages = stage_weights(1536.0, flags=_lib.F_VPOSER)
    res = {}
    for resident in (0, -1):
        eng.set_options(resident_pass=resident)
        caps = []
        for rnd in (9, 140):
            cap = eng.capture_pass(rnd)
            xf, st = eng.fit(x0, stages)
            eng.capture_pass(None)
            assert st['passes']['missed'] == 0 and st['passes']['timed_out'] == 0, st['passes']
            assert st['decoder'] == dict(launches=1, answers_timed_out=0, helpers_gave_up=0), st['decoder']
            caps.append(cap.cpu().numpy())
        res[resident] = dict(x=xf.cpu().numpy(), ncl=st['n_closure'].cpu().numpy(), caps=caps, form=eng.pass_profile()['form'])
    assert res[0]['form'] == 0 and res[-1]['form'] == 3, (res[0]['form'], res[-1]['form'])
    assert np.array_equal(res[0]['x'], res[-1]['x']) and np.array_equal(res[0]['ncl'], res[-1]['ncl'])
    for rnd, a, b in zip((9, 140), res[0]['caps'], res[-1]['caps']):
        have = rnd < res[0]['ncl']
        assert have.any() and np.isfinite(b[have]).all()
        assert np.array_equal(a[have], b[have]), (rnd, np.abs(a[have] - b[have]).max())
    eng.close()
