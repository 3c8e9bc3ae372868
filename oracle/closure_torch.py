"""TEST INFRASTRUCTURE - the CPU baseline leg: a PyTorch-CPU restatement of the reference
closure as an autograd graph ("port" in bench.py's cpu_baseline).

The reference cannot travel to the GPU box, so its cost model is restated here: the same dense
tensor program per closure - all 6890 vertices, dense J_regressor / LSP regressor / skinning
matmuls, per-vertex 4x4 products, autograd backward (reference code/smplx/lbs.py:135-222,
code/smplx/body_models_scale.py:377-403, code/camera.py:93-117, code/utils/fitting.py:290-415) -
driven by the NumPy restatement of LBFGS/run_fitting (oracle/lbfgs_np.py).  float32 like the
reference's default (code/init.py:74-80).  Checked against the float64 oracle in
tests/test_oracle_torch_port.py.  Never imported by the shipped package.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import closure_np as cn
from oracle import lbfgs_np as ln


class TorchClosure:
    def __init__(self, model: dict, cams, gt_xy, w_conf, dtype=torch.float32, vposer=None):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        self.dt = dtype
        self.vt = t(model['v_template'])
        self.S = t(model['shapedirs'])
        self.PD = t(model['posedirs'])
        self.JR = t(model['J_regressor'])
        self.W = t(model['lbs_weights'])
        self.KR = t(model['kp_regressor'])
        self.par = [int(p) for p in model['parents']]
        self.face_ids = torch.tensor(np.asarray(model['face_vertex_ids']), dtype=torch.long)
        self.jmap = torch.tensor(np.asarray(model['joint_map']), dtype=torch.long)
        self.cam_R, self.cam_t, self.cam_f, self.cam_c = (t(a) for a in cams)
        self.gt = t(gt_xy)
        self.w2 = (t(w_conf) ** 2).unsqueeze(-1)
        self.vp = None if vposer is None else {k: t(v) for k, v in vposer.items()}
        self.sgn = t([1.0, -1.0, -1.0, -1.0])
        self.aidx = torch.tensor([52, 55, 9, 12], dtype=torch.long)

    def _rodrigues(self, r):                                       # lbs.py:269-300
        ang = torch.norm(r + 1e-8, dim=1, keepdim=True)
        k = r / ang
        z = torch.zeros_like(k[:, :1])
        K = torch.cat([z, -k[:, 2:3], k[:, 1:2], k[:, 2:3], z, -k[:, 0:1], -k[:, 1:2], k[:, 0:1], z],
                      dim=1).view(-1, 3, 3)
        eye = torch.eye(3, dtype=self.dt).unsqueeze(0)
        return eye + torch.sin(ang).unsqueeze(-1) * K + (1 - torch.cos(ang)).unsqueeze(-1) * torch.bmm(K, K)

    def _vposer(self, z):                                          # VPoser.py:218-232
        F = torch.nn.functional
        h = F.leaky_relu(self.vp['fc1_w'] @ z + self.vp['fc1_b'], 0.2)
        h = F.leaky_relu(self.vp['fc2_w'] @ h + self.vp['fc2_b'], 0.2)
        o = (self.vp['out_w'] @ h + self.vp['out_b']).view(23, 3, 2)
        b1 = F.normalize(o[:, :, 0], dim=1)
        b2 = F.normalize(o[:, :, 1] - (b1 * o[:, :, 1]).sum(1, keepdim=True) * b1, dim=1)
        b3 = torch.cross(b1, b2, dim=1)
        m = torch.stack([b1, b2, b3], dim=1)                       # rows b1,b2,b3 = R^T
        m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
        t0 = 1 + m00 - m11 - m22
        q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
        t1 = 1 - m00 + m11 - m22
        q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
        t2 = 1 - m00 - m11 + m22
        q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
        t3 = 1 + m00 + m11 + m22
        q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
        d2 = m22 < 1e-6
        c0 = (d2 & (m00 > m11)).to(self.dt).unsqueeze(1)
        c1 = (d2 & ~(m00 > m11)).to(self.dt).unsqueeze(1)
        c2 = (~d2 & (m00 < -m11)).to(self.dt).unsqueeze(1)
        c3 = (~d2 & ~(m00 < -m11)).to(self.dt).unsqueeze(1)
        q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
        q = 0.5 * q / torch.sqrt(t0.unsqueeze(1) * c0 + t1.unsqueeze(1) * c1 + t2.unsqueeze(1) * c2 + t3.unsqueeze(1) * c3)
        s2 = (q[:, 1:] ** 2).sum(1)
        s = torch.sqrt(s2)
        tt = 2.0 * torch.where(q[:, 0] < 0, torch.atan2(-s, -q[:, 0]), torch.atan2(s, q[:, 0]))
        kk = torch.where(s2 > 0, tt / s, 2.0 * torch.ones_like(s))
        return (q[:, 1:] * kk.unsqueeze(1)).reshape(69)

    def evaluate(self, x_flat, wts, use_vposer=False):
        """loss (float), grad (numpy) for one problem; x in the oracle's flat layout."""
        lay, _ = cn.param_layout(use_vposer)
        x = torch.tensor(np.asarray(x_flat), dtype=self.dt, requires_grad=True)
        g = lambda n: x[lay[n][0]:lay[n][1]]
        beta, go, tau, sc = g('betas'), g('global_orient'), g('transl'), g('scale')
        bp = self._vposer(g('pose_embedding')) if use_vposer else g('body_pose')
        pose = torch.cat([go, bp]).view(24, 3)
        v_shaped = self.vt + torch.einsum('l,mkl->mk', beta, self.S)
        J = self.JR @ v_shaped
        R = self._rodrigues(pose)
        pf = (R[1:] - torch.eye(3, dtype=self.dt)).reshape(-1)
        v_posed = v_shaped + (pf @ self.PD).view(-1, 3)
        rel = J.clone()
        rel[1:] = J[1:] - J[self.par[1:]]
        Rm = torch.cat([(R[0] * sc).unsqueeze(0), R[1:]], 0)
        M = torch.cat([torch.cat([Rm, rel.unsqueeze(-1)], 2),
                       torch.tensor([0., 0., 0., 1.], dtype=self.dt).expand(24, 1, 4)], 1)
        chain = [M[0]]
        for i in range(1, 24):
            chain.append(chain[self.par[i]] @ M[i])
        G = torch.stack(chain)
        Jh = torch.cat([J, torch.zeros(24, 1, dtype=self.dt)], 1).unsqueeze(-1)
        A = G - torch.nn.functional.pad(G @ Jh, [3, 0])
        T = (self.W @ A.view(24, 16)).view(-1, 4, 4)
        vh = torch.cat([v_posed, torch.ones(v_posed.shape[0], 1, dtype=self.dt)], 1).unsqueeze(-1)
        verts = torch.bmm(T, vh)[:, :3, 0]
        kp = torch.cat([self.KR @ verts, verts[self.face_ids]], 0)[self.jmap] + tau
        p = torch.einsum('vab,kb->vka', self.cam_R, kp) + self.cam_t[:, None, :]
        uv = self.cam_f[:, None, None] * p[..., :2] / p[..., 2:3] + self.cam_c[:, None, :]
        r2 = (self.gt - uv) ** 2
        rho2 = wts['rho'] ** 2
        loss = (self.w2 * (rho2 * r2 / (r2 + rho2))).sum() * wts['data_weight'] ** 2
        wp = wts['body_pose_weight']
        if use_vposer:
            loss = loss + (g('pose_embedding') ** 2).sum() * wp ** 2
        else:
            pp = (bp ** 2).sum() * wp ** 2
            if float(pp.detach()) > 5e4:
                pp = 0.0
            loss = loss + pp + (bp ** 2).sum() * (wp * 4) ** 2
        loss = loss + (beta ** 2).sum() * wts['shape_weight'] ** 2
        ang = (torch.exp(pose.reshape(-1)[3:66][self.aidx] * self.sgn) ** 2).sum() * wts['bending_prior_weight']
        if not (float(ang.detach()) > 1e4 and not use_vposer):
            loss = loss + ang
        loss.backward()
        return float(loss.detach()), x.grad.numpy().astype(np.float64)


def fit_one(tc: TorchClosure, x0, stages, use_vposer=False, max_iter=30, maxiters=30):
    """4-stage fit of one problem on the CPU: returns (x, final_loss, n_closures)."""
    x = np.array(x0, np.float64)
    lay, D = cn.param_layout(use_vposer)
    segs = [lay[k] for k in lay]
    n_cl = 0
    final = None
    for wts in stages:
        opt = ln.LbfgsOracle(x, lambda xx: tc.evaluate(xx, wts, use_vposer), max_iter=max_iter,
                             dtype=np.float32)
        final, _ = ln.run_fitting(opt, maxiters=maxiters, segments=segs)
        x = opt.x.astype(np.float64)
        n_cl += opt.func_evals
    return x, final, n_cl
