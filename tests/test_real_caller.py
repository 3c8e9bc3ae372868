"""CPU (build container only): the reference's OWN, unmodified caller - code/utils/non_linear_solver.py:37-288 -
executed under mvsmplfitting_amd.fitting.patch_reference().

There is no GPU here, so the engine behind the mirror is replaced by tests/stub_engine.StubMvFit (records every
call, numbers from the float64 oracle).  What this proves is the drop-in boundary itself: every attribute / call the
reference caller makes on the patched seams is accepted (`loss.to(device=...)`, `create_optimizer(**all yaml keys)`,
`FittingMonitor(batch_size=..., visualize=..., **kwargs)`, ...), camera / keypoint / weight / parameter plumbing is
right (the fitted parameters equal the unpatched reference's), `optim_type: 'lbfgsls'` (the yaml default,
cfg_files/fit_smpl.yaml:63) runs the reference's own LBFGSLs on our closure, and `'lbfgs_hip'` hands whole stages to
the engine's fit.  On the GPU box the same mirror code drives libmvfit (tests/test_gpu_dropin.py)."""
import numpy as np
import pytest
import torch

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import ref_import as ri
from tests.helpers import body_model

pytestmark = pytest.mark.skipif(not ri.available(), reason='reference tree not mounted')

# the keys of cfg_files/fit_smpl.yaml that main.py forwards as **args (main.py:86) - unknown ones must be swallowed
YAML_KW = dict(
    dataset='offline', joints_to_ign=[-1], prior_folder='priors', result_folder='output', gender='neutral',
    float_dtype='float64', use_pca=True, flat_hand_mean=False, save_meshes=True, num_pca_comps=12,
    body_prior_type='l2', body_tri_idxs=None, df_cone_height=0.0001, penalize_outside=True, max_collisions=128,
    point2plane=False, part_segm_fn='', ign_part_pairs=None, sigma=0.5, data_weights=[1, 1, 1, 1],
    body_pose_prior_weights=[4.04e2, 4.04e2, 57.4e0, 4.78e0], shape_weights=[1e2, 5e1, 1e1, .5e1],
    coll_loss_weights=[0.0, 0.0, 1000., 4500.], use_joints_conf=True, rho=100, lr=1.0, maxiters=30,
    ftol=1e-9, gtol=1e-9, interactive=True, visualize=False, interpenetration=False, use_cuda=False,
    fix_scale=False, fix_shape=False, use_hip=True, model_type='smpllsp', pose_format='lsp14', vposer_ckpt='x')


# named parameters of non_linear_solver (:37-57): they do not travel in its **kwargs
NAMED = ('data_weights', 'body_pose_prior_weights', 'shape_weights', 'coll_loss_weights', 'use_joints_conf', 'rho',
         'interpenetration', 'visualize', 'interactive', 'use_cuda')


def _problem(use_vposer, seed=31, V=4):
    model = body_model()
    cams = syn.make_camera_ring(V)
    vpw = syn.make_vposer_decoder(seed=3, gain=1.0, identity_bias=True) if use_vposer else None
    orc = cn.ClosureOracle(model, np.float64, vposer=None)
    fr = syn.make_frames(1, seed0=seed)
    p = {k: fr[k][0] for k in fr}
    p['use_vposer'] = False
    kp = orc.body(p, want_cache=False)['joints']
    gt, cf = syn.make_observations(kp[None], cams, seed=1)
    return model, cams, vpw, gt[0], cf[0]


def _setting_and_data(model, cams, vpw, gt, cf, use_vposer):
    """What code/init.py:23-205 and data_parser.FittingData hand to non_linear_solver, built from the reference's own
    classes (RefProblem constructs SMPL / cameras / VPoser the way init.py does)."""
    ref = ri.load()
    rp = ri.RefProblem(model, cams, gt, cf, 'float64', use_vposer=use_vposer, vposer_weights=vpw)
    dt = torch.float64
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


def _run(nls, model, cams, vpw, gt, cf, use_vposer, optim_type, n_stages=4, warm_start=None):
    rp, setting, data = _setting_and_data(model, cams, vpw, gt, cf, use_vposer)
    kw = dict(YAML_KW, use_vposer=use_vposer, optim_type=optim_type)
    if warm_start is not None:            # a later frame of a sequence (main.py:76-79: load_init, seq_start False)
        rp.set_flat(warm_start)
        setting['seq_start'] = False
        kw['is_seq'] = True
    for k in ('data_weights', 'body_pose_prior_weights', 'shape_weights', 'coll_loss_weights'):
        kw[k] = kw[k][:n_stages]
    res = nls.non_linear_solver(setting, data, **kw)
    flat = np.concatenate([res[k].reshape(-1) for k in ('betas', 'global_orient', 'transl', 'scale')] +
                          ([res['pose_embedding'].detach().numpy().reshape(-1)] if use_vposer
                           else [res['body_pose'].reshape(-1)]))
    return res, flat


@pytest.fixture()
def seams(monkeypatch):
    """(reference non_linear_solver module, mirror, stub class, reference namespace, patch()) - un-patched on exit."""
    ref = ri.load()
    from utils import non_linear_solver as nls                     # the reference's module, unmodified
    from mvsmplfitting_amd import fitting as mf
    from tests.stub_engine import StubMvFit
    monkeypatch.setattr(mf, 'MvFit', StubMvFit)
    StubMvFit.calls.clear()
    undo = []

    def patch():
        undo.append(mf.patch_reference(ref.fitting, ref.optim_factory))
    yield nls, mf, StubMvFit, ref, patch
    for u in undo:
        u()
    assert ref.optim_factory.create_optimizer is not mf.create_optimizer
    assert ref.fitting.create_loss is not mf.create_loss and mf._reference_create_optimizer is None


@pytest.fixture()
def patched(seams):
    nls, mf, Stub, ref, patch = seams
    patch()
    return nls, mf, Stub, ref


@pytest.mark.parametrize('use_vposer,n_stages', [(False, 4), (True, 2)])
def test_unmodified_non_linear_solver_runs_on_the_patched_seams(seams, use_vposer, n_stages):
    """n_stages = how many of the yaml's four weight sets the caller is given.  With VPoser the last two (weakly
    regularised) stages amplify last-bit differences into different local trajectories (even the
    reference against itself, 1 vs 8 threads), so the exact comparison there covers the first two."""
    nls, mf, Stub, ref, patch = seams
    prob = _problem(use_vposer)
    # (0) the un-patched reference on these inputs
    res_ref, x_ref = _run(nls, *prob, use_vposer, 'lbfgsls', n_stages)
    assert not Stub.calls
    patch()
    # (1) yaml default optimiser: the reference's own LBFGSLs drives our closure from the host
    res_ls, x_ls = _run(nls, *prob, use_vposer, 'lbfgsls', n_stages)
    kinds = [c[0] for c in Stub.calls]
    assert kinds.count('create') == 1 and kinds.count('set_problems') == n_stages and 'fit' not in kinds
    assert kinds.count('closure') > 20
    # the four stages arrived with the caller's weights (non_linear_solver.py:109-124,148-150,177-180)
    seen = []
    for k, d in Stub.calls:
        if k == 'closure' and (not seen or seen[-1] != (d['body_pose_weight'], d['shape_weight'])):
            seen.append((d['body_pose_weight'], d['shape_weight']))
            assert abs(d['data_weight'] - 500 / 1536) < 1e-12
            assert abs(d['bending_prior_weight'] - 3.17 * d['body_pose_weight']) < 1e-9
            assert d['coll_loss_weight'] == 0.0 and d['rho'] == 100.0
    assert seen == [(404.0, 100.0), (404.0, 50.0), (57.4, 10.0), (4.78, 5.0)][:n_stages]
    # (2) opt-in device-resident optimiser: one engine fit per stage
    Stub.calls.clear()
    res_hip, x_hip = _run(nls, *prob, use_vposer, 'lbfgs_hip', n_stages)
    kinds = [c[0] for c in Stub.calls]
    assert kinds.count('fit') == n_stages and 'closure' not in kinds
    # float64 end to end: the oracle equals the reference to 1e-13 per closure, so the three runs walk the same
    # trajectory (they differ only through last-bit differences amplified over a few hundred closures)
    for name, res, x in (('lbfgsls on the HIP closure seam', res_ls, x_ls), ('lbfgs_hip', res_hip, x_hip)):
        assert np.isfinite(res['loss'])
        assert abs(res['loss'] - res_ref['loss']) <= 1e-6 * abs(res_ref['loss']), (name, res['loss'], res_ref['loss'])
        assert np.abs(x - x_ref).max() <= 1e-5, (name, np.abs(x - x_ref).max())


def test_sequence_mode_of_the_unmodified_caller(seams):
    """is_seq (main.py:76-79, init_guess.py:137-166, non_linear_solver.py:158-162): a later frame
    of a sequence starts from the previous frame's result, the caller skips the first two stages and scales the third
    stage's pose weight by 0.15.  All of that is the caller's logic - under the patch it drives the device-resident fit
    with two stages, and the result equals the un-patched reference's."""
    nls, mf, Stub, ref, patch = seams
    prob = _problem(False, seed=44)
    lay, D = cn.param_layout(False)
    prev = np.random.default_rng(8).normal(0, 0.05, D)      # "previous frame": near the rest pose, scale 1
    prev[lay['scale'][0]] = 1.0
    res_ref, x_ref = _run(nls, *prob, False, 'lbfgsls', warm_start=prev)
    patch()
    res_hip, x_hip = _run(nls, *prob, False, 'lbfgs_hip', warm_start=prev)
    fits = [d for k, d in Stub.calls if k == 'fit']
    assert len(fits) == 2                                                         # stages 0 and 1 were skipped
    res_ls, x_ls = _run(nls, *prob, False, 'lbfgsls', warm_start=prev)
    w3 = [d['body_pose_weight'] for k, d in Stub.calls if k == 'closure']
    assert abs(w3[0] - 57.4 * 0.15) < 1e-6 and abs(w3[-1] - 4.78) < 1e-9          # :162 and the last stage
    for name, res, x in (('lbfgs_hip', res_hip, x_hip), ('lbfgsls on the HIP closure seam', res_ls, x_ls)):
        assert abs(res['loss'] - res_ref['loss']) <= 1e-6 * abs(res_ref['loss']), (name, res['loss'], res_ref['loss'])
        assert np.abs(x - x_ref).max() <= 1e-5, (name, np.abs(x - x_ref).max())


def test_seam_details_the_caller_relies_on(patched):
    nls, mf, Stub, ref = patched
    loss = ref.fitting.create_loss(loss_type='smplify', joint_weights=None, rho=100, use_joints_conf=True, vposer=None,
                                   pose_embedding=None, body_pose_prior=None, shape_prior=None, angle_prior=None,
                                   interpenetration=False, pen_distance=None, search_tree=None,
                                   tri_filtering_module=None, dtype=torch.float32, use_3d=False,
                                   **{k: v for k, v in YAML_KW.items() if k not in NAMED})
    assert loss.to(device=torch.device('cpu')) is loss             # non_linear_solver.py:143
    assert isinstance(loss, torch.nn.Module)
    loss.reset_loss_weights({'data_weight': torch.tensor(0.3), 'body_pose_weight': 2.0, 'no_such_weight': 1.0})
    assert abs(loss.data_weight - 0.3) < 1e-7 and loss.body_pose_weight == 2.0 and not hasattr(loss, 'no_such_weight')
    with pytest.raises(mf.MvFitError):
        loss(None)
    with pytest.raises(ValueError):
        ref.fitting.create_loss(loss_type='nope')
    p = [torch.nn.Parameter(torch.zeros(3))]
    KW = {k: v for k, v in YAML_KW.items() if k not in NAMED}
    # every optim_type other than 'lbfgs_hip' is built by the reference's own factory (optim_factory.py:27-65)
    opt, cg = ref.optim_factory.create_optimizer(p, **dict(KW, optim_type='lbfgsls'))
    assert type(opt).__module__.endswith('lbfgs_ls') and cg is False
    opt, _ = ref.optim_factory.create_optimizer(p, **dict(KW, optim_type='adam'))
    assert isinstance(opt, torch.optim.Adam)
    with pytest.raises(TypeError):                                  # the reference's rmsprop branch passes epsilon= (:53-58)
        ref.optim_factory.create_optimizer(p, **dict(KW, optim_type='rmsprop'))
    with pytest.raises(ValueError, match='not supported'):
        ref.optim_factory.create_optimizer(p, **dict(KW, optim_type='nope'))
    opt, cg = ref.optim_factory.create_optimizer(p, **dict(KW, optim_type='lbfgs_hip'))
    assert isinstance(opt, mf.LBFGSHip) and opt.max_iter == 30 and opt.lr == 1.0 and cg is False
    mon = ref.fitting.FittingMonitor(batch_size=1, visualize=False,
                                     **{k: v for k, v in YAML_KW.items() if k not in NAMED})
    assert (mon.maxiters, mon.ftol, mon.gtol) == (30, 1e-9, 1e-9)


def test_standalone_factory_matches_reference_errors():
    """Without patch_reference (GPU box: no reference tree) the torch optimisers are built with the reference's
    arguments and 'lbfgsls' is refused with the reference's exception type."""
    from mvsmplfitting_amd import fitting as mf
    assert mf._reference_create_optimizer is None
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert isinstance(mf.create_optimizer(p, optim_type='sgd', lr=0.1)[0], torch.optim.SGD)
    assert isinstance(mf.create_optimizer(p, optim_type='lbfgs', lr=1.0, maxiters=7)[0], torch.optim.LBFGS)
    with pytest.raises(TypeError):
        mf.create_optimizer(p, optim_type='rmsprop')
    with pytest.raises(ValueError):
        mf.create_optimizer(p, optim_type='lbfgsls')
    with pytest.raises(ValueError):
        mf.create_optimizer(p, optim_type='nope')
