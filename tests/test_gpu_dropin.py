"""GPU: the host-side mirror of the reference's seams (mvsmplfitting_amd/fitting.py) driven the way
reference code/utils/non_linear_solver.py:127-203 drives them - create_loss, per-stage create_optimizer +
reset_loss_weights + create_fitting_closure + run_fitting - and checked against the reference's own
golden outputs (closure values, 4-stage fits)."""
import os
import types

import numpy as np
import pytest
import torch

from mvsmplfitting_amd import fitting as mf
from mvsmplfitting_amd.engine import YAML_POSE_W, YAML_SHAPE_W
from tests.helpers import GOLD, body_model, load_case, oracle_for

pytestmark = pytest.mark.gpu


def _cameras(g):
    cams = []
    for v in range(g['cam_R'].shape[0]):
        cams.append(types.SimpleNamespace(
            rotation=torch.tensor(g['cam_R'][v]).unsqueeze(0), translation=torch.tensor(g['cam_t'][v]).unsqueeze(0),
            focal_length_x=torch.tensor([float(g['cam_f'][v])]), center=torch.tensor(g['cam_c'][v]).unsqueeze(0)))
    return cams


def _frame_inputs(g, b):
    gt_joints = torch.tensor(g['gt_xy'][b][:, None])                       # [V,1,17,2] (non_linear_solver.py:77-78)
    joints_conf = [torch.tensor(g['conf'][b][v][None]) for v in range(g['conf'].shape[1])]
    return gt_joints, joints_conf, torch.ones(1, 17)


def test_closure_seam_matches_reference_golden():
    cfg, g, model, vpw, gmm, wts, cams = load_case('l2_s0_v8')
    bm = mf.BodyModel(model)
    loss = mf.create_loss(loss_type='smplify', rho=wts['rho'], body_pose_prior=None, interpenetration=False)
    loss.reset_loss_weights({k: torch.tensor(v) for k, v in wts.items() if k != 'rho'})
    monitor = mf.FittingMonitor(maxiters=30, ftol=1e-9, gtol=1e-9)
    for b in range(2):
        x = g['x'][b]
        bm.reset_params(betas=x[0:10], global_orient=x[10:13], body_pose=x[13:82], transl=x[82:85], scale=x[85:86])
        params = [p for p in bm.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=0.0)                               # any optimiser with zero_grad()
        gt_joints, joints_conf, jw = _frame_inputs(g, b)
        closure = monitor.create_fitting_closure(opt, bm, camera=_cameras(g), gt_joints=gt_joints, loss=loss,
                                                 joints_conf=joints_conf, joint_weights=jw,
                                                 return_verts=True, return_full_pose=True, use_vposer=False)
        val = closure(backward=True)
        assert val.dim() == 0 and val.is_cuda and not torch.isnan(val)
        assert abs(float(val) - g['loss64'][b]) <= 1e-5 * abs(g['loss64'][b])
        grad = torch.cat([p.grad.reshape(-1) for p in params]).cpu().numpy().astype(np.float64)
        assert np.abs(grad - g['grad64'][b]).max() <= 2e-4 * np.abs(g['grad64'][b]).max()
        # forward-only call leaves .grad alone (fitting.py:190)
        before = [p.grad.clone() for p in params]
        closure(backward=False)
        assert all(torch.equal(a, p.grad) for a, p in zip(before, params))
    with pytest.raises(ValueError):
        mf.create_loss(loss_type='nope')
    with pytest.raises(ValueError):
        mf.create_optimizer(params, optim_type='nope')


def test_stage_loop_like_non_linear_solver():
    g = dict(np.load(os.path.join(GOLD, 'fit_l2.npz')))
    model = body_model()
    H = 1536.0
    for b in range(g['x0'].shape[0]):
        bm = mf.BodyModel(model)
        x0 = g['x0'][b]
        bm.reset_params(betas=x0[0:10], global_orient=x0[10:13], body_pose=x0[13:82], transl=x0[82:85], scale=x0[85:86])
        loss = mf.create_loss(loss_type='smplify', rho=100, interpenetration=False)
        monitor = mf.FittingMonitor(maxiters=30, ftol=1e-9, gtol=1e-9)
        gt_joints, joints_conf, jw = _frame_inputs(g, b)
        final = None
        for s in range(4):                                                   # non_linear_solver.py:156-203
            final_params = [p for p in bm.parameters() if p.requires_grad]
            opt, create_graph = mf.create_optimizer(final_params, optim_type='lbfgs_hip', lr=1.0, maxiters=30)
            opt.zero_grad()
            w = dict(data_weight=500.0 / H, body_pose_weight=YAML_POSE_W[s], shape_weight=YAML_SHAPE_W[s])
            w['bending_prior_weight'] = 3.17 * w['body_pose_weight']
            loss.reset_loss_weights(w)
            closure = monitor.create_fitting_closure(opt, bm, camera=_cameras(g), gt_joints=gt_joints, loss=loss,
                                                     joints_conf=joints_conf, joint_weights=jw,
                                                     create_graph=create_graph, use_vposer=False)
            final = monitor.run_fitting(opt, closure, final_params, bm, use_vposer=False)
        assert final is not None and np.isfinite(final)
        assert final <= 1.25 * g['final'][b] + 1.0, (final, g['final'][b])
        # the fitted values are left in the torch Parameters (non_linear_solver.py:284-287)
        chk = float(closure(backward=False))
        assert chk <= final * (1 + 1e-3) + 1e-3


def test_closure_seam_with_interpenetration():
    """create_loss(interpenetration=True) + coll_loss_weight > 0 through the reference's seams: the SDF term
    of fitting.py:352-393 as wired (first triangle only, grid 128)."""
    from oracle import sdf_term_np as st
    cfg, g, model, vpw, gmm, wts, cams = load_case('l2_s3_v6')
    orc = oracle_for(model, vpw, gmm)
    bm = mf.BodyModel(model)
    loss = mf.create_loss(loss_type='smplify', rho=wts['rho'], body_pose_prior=None)     # interpenetration defaults to True
    assert loss.interpenetration
    cw = 40.0
    loss.reset_loss_weights({k: torch.tensor(v) for k, v in dict(wts, coll_loss_weight=cw).items() if k != 'rho'})
    monitor = mf.FittingMonitor(maxiters=30, ftol=1e-9, gtol=1e-9)
    x = g['x'][0]
    bm.reset_params(betas=x[0:10], global_orient=x[10:13], body_pose=x[13:82], transl=x[82:85], scale=x[85:86])
    params = [p for p in bm.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0)
    gt_joints, joints_conf, jw = _frame_inputs(g, 0)
    closure = monitor.create_fitting_closure(opt, bm, camera=_cameras(g), gt_joints=gt_joints, loss=loss,
                                             joints_conf=joints_conf, joint_weights=jw, use_vposer=False)
    val = float(closure(backward=True))
    verts = closure.eng.closure(closure.pack(), loss.weights(closure.flags), want_grad=False, want_verts=True)['verts']
    pen, g_sdf, aux = st.sdf_term(verts[0].cpu().numpy().astype(np.float64), model['faces'], cw, 1, 128)
    assert aux['S'] > 0
    Lr, gr, _ = orc.closure(x, cams, g['gt_xy'][0], g['conf'][0], dict(wts, coll_loss_weight=0.0), g_verts_extra=g_sdf)
    assert abs(val - (Lr + pen)) <= 1e-5 * abs(Lr + pen)
    grad = torch.cat([p.grad.reshape(-1) for p in params]).cpu().numpy().astype(np.float64)
    assert np.abs(grad - gr).max() <= 2e-4 * np.abs(gr).max()
    # weight 0 -> the term is off (fitting.py:354)
    loss.reset_loss_weights({'coll_loss_weight': 0.0})
    assert abs(float(closure(backward=False)) - Lr) <= 1e-5 * abs(Lr)
