"""GPU: the reference's OWN, unmodified caller - code/utils/non_linear_solver.py:37-288 - executed under
mvsmplfitting_amd.fitting.patch_reference() against the REAL libmvfit.so.

tests/test_real_caller.py runs the same caller in the build container against a recording stub (no GPU there);
tests/test_gpu_dropin.py runs the real library under a restated driving sequence (no reference on the GPU box).  This
file is the product of the two - SURVEY 8(b)'s drop-in claim executed as stated: the reference's modules come from
oracle/_ref/reference_stage.tgz (`make -C oracle stage`: the reference's own files, archived where they lie, git-ignored,
shipped with the snapshot like libsdf_ref.so - test infrastructure, never imported by the product), the un-patched
reference runs on the host cores IN THE SAME PROCESS as the yard-stick, and

  * every closure value / gradient the patched caller saw from libmvfit (`'lbfgsls'`: the reference's own LBFGSLs drives the
    HIP closure from the host) is re-evaluated by the reference's own `create_fitting_closure` closure at the same point
    with the same stage weights: 1e-5 relative on the loss (north_star), 2e-4 of the gradient's largest entry;
  * the fits (`'lbfgsls'` and the device-resident `'lbfgs_hip'`, with / without VPoser, plain and `is_seq`) end inside the
    spread of the reference's own float32 / float64 fits of the same frame.
"""
import numpy as np
import pytest
import torch

from oracle import closure_np as cn
from oracle import ref_import as ri
from tests.test_real_caller import YAML_KW, _problem

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ri.available(), reason='no reference: neither /root/reference nor oracle/_ref/reference_stage.tgz (make -C oracle stage)')]

WEIGHT_KEYS = ('data_weight', 'body_pose_weight', 'shape_weight', 'bending_prior_weight', 'coll_loss_weight')


def _setting_and_data(model, cams, vpw, gt, cf, use_vposer, dtype):
    """What code/init.py:23-205 and data_parser.FittingData hand to non_linear_solver, from the reference's own classes."""
    ref = ri.load()
    rp = ri.RefProblem(model, cams, gt, cf, dtype, use_vposer=use_vposer, vposer_weights=vpw)
    dt = rp.dt
    setting = dict(views=cams[0].shape[0], device=torch.device('cpu'), dtype=dt, vposer=rp.vposer,
                   joints_weight=rp.joint_weights, model=rp.smpl, camera=rp.cameras,
                   pose_embedding=rp.pose_embedding, seq_start=True, adjustment=False,
                   body_pose_prior=ref.prior.create_prior('l2', dtype=dt),
                   shape_prior=ref.prior.create_prior('l2', dtype=dt),
                   angle_prior=ref.prior.create_prior('angle', dtype=dt))
    V = cams[0].shape[0]
    kps = np.concatenate([gt, cf[..., None]], -1)[:, None]            # [V, P=1, 17, 3]  (data_parser.py:42-90)
    data = {'keypoints': kps.astype(np.float64), '3d_joint': None, 'img': [np.zeros((1536, 2048, 3), np.uint8)] * V,
            'img_path': ['x.jpg'] * V}
    return rp, setting, data


def _run(nls, prob, use_vposer, optim_type, dtype, n_stages=4, warm_start=None, start=None):
    rp, setting, data = _setting_and_data(*prob, use_vposer, dtype)
    kw = dict(YAML_KW, use_vposer=use_vposer, optim_type=optim_type, float_dtype=dtype)
    if start is not None:                 # another start point of the same (first-frame) fit
        rp.set_flat(start)
    if warm_start is not None:            # a later frame of a sequence (main.py:76-79: load_init, seq_start False)
        rp.set_flat(warm_start)
        setting['seq_start'] = False
        kw['is_seq'] = True
    for k in ('data_weights', 'body_pose_prior_weights', 'shape_weights', 'coll_loss_weights'):
        kw[k] = kw[k][:n_stages]
    res = nls.non_linear_solver(setting, data, **kw)
    return res, rp


@pytest.fixture()
def seams():
    ref = ri.load()
    from utils import non_linear_solver as nls                     # the reference's module, unmodified
    from mvsmplfitting_amd import fitting as mf
    undo = []

    def patch():
        undo.append(mf.patch_reference(ref.fitting, ref.optim_factory))

    def unpatch():
        while undo:
            undo.pop()()
    yield nls, mf, ref, patch, unpatch
    unpatch()
    assert ref.fitting.create_loss is not mf.create_loss and mf._reference_create_optimizer is None


def _recording(mf, monkeypatch, log):
    """Record what libmvfit answered at every closure call of the patched caller: parameter values by name, the stage
    weights in force, loss and gradients."""
    orig = mf._HipClosure.__call__

    def call(self, backward=True):
        val = orig(self, backward)
        if backward:
            log.append(dict(
                params={k: p.detach().cpu().numpy().astype(np.float64).copy() for k, p in self.params.items()},
                grads={k: p.grad.detach().cpu().numpy().astype(np.float64).copy() for k, p in self.params.items()
                       if p.requires_grad and p.grad is not None},
                weights={k: float(getattr(self.loss, k)) for k in WEIGHT_KEYS}, loss=float(val)))
        return val
    monkeypatch.setattr(mf._HipClosure, '__call__', call)


def _reference_closure_at(rp, rec):
    """The reference's own fitting_func (fitting.py:162-203, built by ITS create_fitting_closure) at a recorded point."""
    named = dict(rp.smpl.named_parameters())
    if rp.pose_embedding is not None:
        named['pose_embedding'] = rp.pose_embedding
    with torch.no_grad():
        for k, v in rec['params'].items():
            named[k].copy_(torch.tensor(v, dtype=rp.dt).view_as(named[k]))
    rp.set_weights(rec['weights'])
    closure = rp.make_closure(rp.make_optimizer())
    loss = float(closure(backward=True))
    return loss, {k: named[k].grad.detach().numpy().astype(np.float64) for k in rec['grads']}


@pytest.mark.parametrize('use_vposer,n_stages', [(False, 4), (True, 2), (True, 4)])
def test_unmodified_caller_drives_libmvfit(seams, monkeypatch, use_vposer, n_stages):
    nls, mf, ref, patch, unpatch = seams
    prob = _problem(use_vposer)
    # (0) the reference itself, un-patched, on this box's host cores: float64 and float32
    ref_loss = {}
    for dtype in ('float64', 'float32'):
        res, _ = _run(nls, prob, use_vposer, 'lbfgsls', dtype, n_stages)
        ref_loss[dtype] = float(res['loss'])
    worst = max(ref_loss.values())
    # an un-patched reference problem for the closure-level comparison (its loss / monitor objects are the reference's)
    # ... and its own float32 closure at the same point: the yard-stick for the gradient (the gradient is the small
    # difference of large data / prior terms; float32 rounding leaves ~0.1 absolute in the reference as on the device).
    # Both are built BEFORE patch(): their loss / monitor objects must be the reference's, not the mirror's.
    rp64, _, _ = _setting_and_data(*prob, use_vposer, 'float64')
    rp32, _, _ = _setting_and_data(*prob, use_vposer, 'float32')
    assert type(rp32.monitor).__module__.endswith('utils.fitting') and type(rp32.loss).__module__.endswith('utils.fitting')
    patch()
    # (1) yaml default optimiser: the reference's own LBFGSLs drives the HIP closure from the host
    log = []
    _recording(mf, monkeypatch, log)
    res_ls, _ = _run(nls, prob, use_vposer, 'lbfgsls', 'float32', n_stages)
    assert len(log) > 20 * n_stages
    seen = []
    for r in log:
        w = (r['weights']['body_pose_weight'], r['weights']['shape_weight'])
        if not seen or seen[-1] != w:
            seen.append(w)
    want = [(404.0, 100.0), (404.0, 50.0), (57.4, 10.0), (4.78, 5.0)][:n_stages]      # the caller's float32 weight tensors
    assert len(seen) == n_stages and np.allclose(seen, want, rtol=1e-6), seen
    # closure level: every 7th recorded call + the first call of each stage, against the reference's own closure
    firsts = [next(i for i, r in enumerate(log) if (r['weights']['body_pose_weight'], r['weights']['shape_weight']) == w) for w in seen]
    picks = sorted(set(firsts) | set(range(0, len(log), 7)))
    worst_l, worst_g, worst_info = 0.0, 0.0, None
    for i in picks:
        rec = log[i]
        l_ref, g_ref = _reference_closure_at(rp64, rec)
        _, g_r32 = _reference_closure_at(rp32, rec)
        worst_l = max(worst_l, abs(rec['loss'] - l_ref) / abs(l_ref))
        gmax = max(np.abs(v).max() for v in g_ref.values())
        for k, v in rec['grads'].items():
            err = np.abs(v - g_ref[k]).max()
            err32 = max(np.abs(g_r32[q] - g_ref[q]).max() for q in g_ref)
            excess = err / (2e-4 * gmax + 4.0 * err32)
            if excess > worst_g:
                worst_g, worst_info = excess, dict(call=i, of=len(log), param=k, err=err, gmax=gmax, reference_float32_err=err32)
    assert worst_l <= 1e-5, worst_l                     # north_star: 1e-5 on the scalar loss
    # gradient: 2e-4 of its largest entry (the bound of the golden closure tests) + 4 x what the reference's own float32
    # closure deviates from its float64 one at that point
    assert worst_g <= 1.0, worst_info
    # (2) opt-in device-resident optimiser: whole stages inside mvfit_fit
    res_hip, _ = _run(nls, prob, use_vposer, 'lbfgs_hip', 'float32', n_stages)
    unpatch()
    # fit level: inside the spread of the reference's own float32 / float64 fits.  VPoser's weakly regularised last stages
    # amplify last-bit differences into other local trajectories (the reference against itself, too): for that
    # configuration the yard-stick is the reference's own float32 spread over 24 starts perturbed in the last bits
    # (oracle/make_golden_real_caller_spread.py -> tests/golden/real_caller_vposer_spread.npz: 1131.8 ... 1293.7, five of them
    # above 1160) - the device fits must not end worse than the worst of those
    bound = 1.05 * worst
    if use_vposer and n_stages == 4:
        from tests.helpers import GOLD
        import os
        bound = 1.02 * float(np.load(os.path.join(GOLD, 'real_caller_vposer_spread.npz'))['final32'].max())
    for name, res in (('lbfgsls on the HIP closure', res_ls), ('lbfgs_hip', res_hip)):
        assert np.isfinite(res['loss'])
        assert float(res['loss']) <= bound, (name, float(res['loss']), ref_loss, bound)
    print('real caller: use_vposer=%s stages=%d reference fp64 / fp32 %.4f / %.4f, lbfgsls on libmvfit %.4f, lbfgs_hip %.4f; '
          'closure level over %d points: loss %.1e rel, gradient at %.2f of its bound (worst: %s)'
          % (use_vposer, n_stages, ref_loss['float64'], ref_loss['float32'], float(res_ls['loss']), float(res_hip['loss']),
             len(picks), worst_l, worst_g, worst_info))


def test_sequence_mode_of_the_unmodified_caller_on_libmvfit(seams):
    """is_seq (main.py:76-79, init_guess.py:137-166, non_linear_solver.py:158-162): warm start from the previous frame, the
    caller skips the first two stages and scales the third stage's pose weight by 0.15 - its own logic, on libmvfit."""
    nls, mf, ref, patch, unpatch = seams
    prob = _problem(False, seed=44)
    lay, D = cn.param_layout(False)
    prev = np.random.default_rng(8).normal(0, 0.05, D)      # "previous frame": near the rest pose, scale 1
    prev[lay['scale'][0]] = 1.0
    ref_loss = [float(_run(nls, prob, False, 'lbfgsls', dt, warm_start=prev)[0]['loss']) for dt in ('float64', 'float32')]
    patch()
    for optim in ('lbfgs_hip', 'lbfgsls'):
        res, _ = _run(nls, prob, False, optim, 'float32', warm_start=prev)
        assert np.isfinite(res['loss']) and float(res['loss']) <= 1.05 * max(ref_loss), (optim, float(res['loss']), ref_loss)
    unpatch()
