"""TEST INFRASTRUCTURE - CPU oracle for the multi-view SMPL fitting closure.

An independent NumPy restatement (default float64) of ONE evaluation of the
reference closure ``fitting_func`` (reference code/utils/fitting.py:162-203):
forward *and* hand-derived reverse-mode adjoint.  It is the checker for the HIP path;
nothing in the shipped package may import it (only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do).

Parity pin: the reference has no golden vectors of its own, so this
file is pinned against the *reference itself* imported in the build container
(oracle/ref_import.py; tests/test_oracle_vs_reference.py) and against the golden
vectors that import produced (tests/golden/*.npz, written by oracle/make_golden.py).

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import numpy as np

ANGLE_IDX = np.array([52, 55, 9, 12])            # prior.py:63 minus 3 (prior.py:87)
ANGLE_SGN = np.array([1.0, -1.0, -1.0, -1.0])     # prior.py:67

# flag values shared with include/mvfit.h
PRIOR_L2 = 0
PRIOR_GMM = 1


def param_layout(use_vposer: bool):
    """Flat parameter order = ``final_params`` of the reference
    (non_linear_solver.py:164-170: model.parameters() in registration order
    betas, global_orient, [body_pose], transl, scale  (body_models_scale.py:202-268),
    then pose_embedding)."""
    if use_vposer:
        names = [('betas', 10), ('global_orient', 3), ('transl', 3), ('scale', 1),
                 ('pose_embedding', 32)]
    else:
        names = [('betas', 10), ('global_orient', 3), ('body_pose', 69), ('transl', 3),
                 ('scale', 1)]
    off, out = 0, {}
    for n, k in names:
        out[n] = (off, off + k)
        off += k
    return out, off


def pack(params: dict, use_vposer: bool, dtype=np.float64):
    lay, D = param_layout(use_vposer)
    x = np.zeros(D, dtype)
    for n, (a, b) in lay.items():
        x[a:b] = np.asarray(params[n], dtype).reshape(-1)
    return x


def unpack(x, use_vposer: bool):
    lay, _ = param_layout(use_vposer)
    return {n: x[a:b] for n, (a, b) in lay.items()}


# ------------------------------------------------------------------ Rodrigues
def rodrigues(r):
    """lbs.py:269-300 - note the 1e-8 added per component *inside* the norm."""
    e = r + 1e-8
    a = np.sqrt((e * e).sum())
    k = r / a
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]], r.dtype)
    R = np.eye(3, dtype=r.dtype) + np.sin(a) * K + (1.0 - np.cos(a)) * (K @ K)
    return R, (a, k, K, e)


def rodrigues_bwd(gR, r, cache):
    a, k, K, e = cache
    KK = K @ K
    g_a = np.cos(a) * (gR * K).sum() + np.sin(a) * (gR * KK).sum()
    g_K = np.sin(a) * gR + (1.0 - np.cos(a)) * (gR @ K.T + K.T @ gR)
    g_k = np.array([g_K[2, 1] - g_K[1, 2], g_K[0, 2] - g_K[2, 0], g_K[1, 0] - g_K[0, 1]])
    g_a = g_a - (g_k * r).sum() / (a * a)
    return g_k / a + g_a * e / a


# ------------------------------------------------------------------ VPoser decoder
def _lrelu(x):
    return np.where(x > 0, x, 0.2 * x)


def vposer_decode(z, vp):
    """VPoser.decode(z, 'aa') (VPoser.py:218-232, 165-174, 263-273, 29-156)."""
    pre1 = vp['fc1_w'] @ z + vp['fc1_b']
    h1 = _lrelu(pre1)
    pre2 = vp['fc2_w'] @ h1 + vp['fc2_b']
    h2 = _lrelu(pre2)
    o = (vp['out_w'] @ h2 + vp['out_b']).reshape(23, 3, 2)
    aa = np.zeros((23, 3), z.dtype)
    caches = []
    for j in range(23):
        a1, a2 = o[j, :, 0], o[j, :, 1]
        n1 = max(np.sqrt((a1 * a1).sum()), 1e-12)            # F.normalize eps
        b1 = a1 / n1
        d = (b1 * a2).sum()
        u = a2 - d * b1
        n2 = max(np.sqrt((u * u).sum()), 1e-12)
        b2 = u / n2
        b3 = np.cross(b1, b2)
        m = np.stack([b1, b2, b3], axis=0)       # m = R^T : rows are b1,b2,b3 (VPoser.py:62)
        if m[2, 2] < 1e-6:
            if m[0, 0] > m[1, 1]:
                case = 0
                t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
                q = np.array([m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]])
            else:
                case = 1
                t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
                q = np.array([m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]])
        else:
            if m[0, 0] < -m[1, 1]:
                case = 2
                t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
                q = np.array([m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t])
            else:
                case = 3
                t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
                q = np.array([t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]])
        qraw = q
        q = 0.5 * qraw / np.sqrt(t)
        s2 = q[1] ** 2 + q[2] ** 2 + q[3] ** 2
        s = np.sqrt(s2)
        c = q[0]
        tt = 2.0 * (np.arctan2(-s, -c) if c < 0 else np.arctan2(s, c))
        kk = tt / s if s2 > 0 else 2.0
        aa[j] = q[1:] * kk
        caches.append((a1, a2, n1, b1, d, u, n2, b2, b3, m, case, t, qraw, q, s2, s, c, tt, kk))
    return aa.reshape(69), (z, pre1, h1, pre2, h2, caches)


def vposer_decode_bwd(g_bp, vp, cache):
    z, pre1, h1, pre2, h2, caches = cache
    g_bp = g_bp.reshape(23, 3)
    g_o = np.zeros((23, 3, 2), z.dtype)
    for j in range(23):
        (a1, a2, n1, b1, d, u, n2, b2, b3, m, case, t, qraw, q, s2, s, c, tt, kk) = caches[j]
        g_aa = g_bp[j]
        g_q = np.zeros(4, z.dtype)
        g_q[1:] = g_aa * kk
        g_k = (g_aa * q[1:]).sum()
        if s2 > 0:
            g_tt = g_k / s
            den = s2 + c * c
            g_s = -g_k * tt / s2 + g_tt * 2.0 * c / den
            g_c = g_tt * (-2.0 * s / den)
            g_q[1:] += 2.0 * q[1:] * (g_s / (2.0 * s))
            g_q[0] += g_c
        # q = 0.5 qraw / sqrt(t), and t is itself one of qraw's components
        g_qraw = 0.5 * g_q / np.sqrt(t)
        g_t = (g_q * qraw).sum() * 0.5 * (-0.5) * t ** (-1.5)
        g_m = np.zeros((3, 3), z.dtype)

        def acc(i, jj, val):
            g_m[i, jj] += val
        if case == 0:
            g_t += g_qraw[1]
            acc(1, 2, g_qraw[0]); acc(2, 1, -g_qraw[0])
            acc(0, 1, g_qraw[2]); acc(1, 0, g_qraw[2])
            acc(2, 0, g_qraw[3]); acc(0, 2, g_qraw[3])
            acc(0, 0, g_t); acc(1, 1, -g_t); acc(2, 2, -g_t)
        elif case == 1:
            g_t += g_qraw[2]
            acc(2, 0, g_qraw[0]); acc(0, 2, -g_qraw[0])
            acc(0, 1, g_qraw[1]); acc(1, 0, g_qraw[1])
            acc(1, 2, g_qraw[3]); acc(2, 1, g_qraw[3])
            acc(0, 0, -g_t); acc(1, 1, g_t); acc(2, 2, -g_t)
        elif case == 2:
            g_t += g_qraw[3]
            acc(0, 1, g_qraw[0]); acc(1, 0, -g_qraw[0])
            acc(2, 0, g_qraw[1]); acc(0, 2, g_qraw[1])
            acc(1, 2, g_qraw[2]); acc(2, 1, g_qraw[2])
            acc(0, 0, -g_t); acc(1, 1, -g_t); acc(2, 2, g_t)
        else:
            g_t += g_qraw[0]
            acc(1, 2, g_qraw[1]); acc(2, 1, -g_qraw[1])
            acc(2, 0, g_qraw[2]); acc(0, 2, -g_qraw[2])
            acc(0, 1, g_qraw[3]); acc(1, 0, -g_qraw[3])
            acc(0, 0, g_t); acc(1, 1, g_t); acc(2, 2, g_t)
        g_b1, g_b2, g_b3 = g_m[0].copy(), g_m[1].copy(), g_m[2].copy()
        # b3 = b1 x b2
        g_b1 += np.cross(b2, g_b3)
        g_b2 += np.cross(g_b3, b1)
        # b2 = u / n2
        g_u = (g_b2 - b2 * (b2 * g_b2).sum()) / n2
        # u = a2 - d b1 ; d = b1.a2
        g_a2 = g_u.copy()
        g_d = -(g_u * b1).sum()
        g_b1 += -d * g_u + g_d * a2
        g_a2 += g_d * b1
        g_a1 = (g_b1 - b1 * (b1 * g_b1).sum()) / n1
        g_o[j, :, 0] = g_a1
        g_o[j, :, 1] = g_a2
    g_o = g_o.reshape(138)
    g_h2 = vp['out_w'].T @ g_o
    g_pre2 = g_h2 * np.where(pre2 > 0, 1.0, 0.2)
    g_h1 = vp['fc2_w'].T @ g_pre2
    g_pre1 = g_h1 * np.where(pre1 > 0, 1.0, 0.2)
    return vp['fc1_w'].T @ g_pre1


# ------------------------------------------------------------------ the closure
class ClosureOracle:
    """One (subject, frame) problem at a time; constants are the float32 model arrays
    upcast to ``dtype`` (the reference rounds them to float32 too, smplx/utils.py:36-39).
    """

    def __init__(self, model: dict, dtype=np.float64, vposer: dict | None = None,
                 gmm=None):
        c = lambda a: np.asarray(a, dtype)
        self.dtype = dtype
        self.vt = c(model['v_template'])
        self.S = c(model['shapedirs'])
        self.PD = c(model['posedirs'])
        self.JR = c(model['J_regressor'])
        self.W = c(model['lbs_weights'])
        self.KR = c(model['kp_regressor'])
        self.par = np.asarray(model['parents'], np.int64)
        self.face_ids = np.asarray(model['face_vertex_ids'], np.int64)
        self.jmap = np.asarray(model['joint_map'], np.int64)
        self.vp = None if vposer is None else {k: c(v) for k, v in vposer.items()}
        self.gmm = None if gmm is None else tuple(c(a) for a in gmm)  # means, precisions, nll_w
        # 17 x 6890 selection matrix (LSP rows + one-hot face vertices, remapped)
        sel19 = np.zeros((19, self.vt.shape[0]), dtype)
        sel19[:14] = self.KR
        sel19[14 + np.arange(5), self.face_ids] = 1.0
        self.Ksel = sel19[self.jmap]

    # -- SMPL forward: body_models_scale.py:327-412 + lbs.py:135-222
    def body(self, p: dict, want_cache=True):
        dt = self.dtype
        beta = np.asarray(p['betas'], dt).reshape(10)
        cache_vp = None
        if 'pose_embedding' in p and p.get('pose_embedding') is not None and self.vp is not None \
                and p.get('use_vposer', True):
            body_pose, cache_vp = vposer_decode(np.asarray(p['pose_embedding'], dt).reshape(32),
                                                self.vp)
        else:
            body_pose = np.asarray(p['body_pose'], dt).reshape(69)
        theta = np.concatenate([np.asarray(p['global_orient'], dt).reshape(3), body_pose]).reshape(24, 3)
        tau = np.asarray(p['transl'], dt).reshape(3)
        s = dt(np.asarray(p['scale']).reshape(()))

        v_shaped = self.vt + self.S @ beta                                    # lbs.py:179
        J = self.JR @ v_shaped                                                # lbs.py:183
        R = np.zeros((24, 3, 3), dt)
        rc = []
        for i in range(24):
            R[i], c_ = rodrigues(theta[i])
            rc.append(c_)
        pf = (R[1:] - np.eye(3, dtype=dt)).reshape(207)                       # lbs.py:192
        v_posed = v_shaped + (pf @ self.PD).reshape(-1, 3)                    # lbs.py:194-203
        # kinematic chain: lbs.py:316-370
        Rm = R.copy()
        Rm[0] = s * R[0]                                                      # lbs.py:348
        tm = J.copy()
        tm[1:] = J[1:] - J[self.par[1:]]
        Gr = np.zeros((24, 3, 3), dt)
        Gt = np.zeros((24, 3), dt)
        Gr[0], Gt[0] = Rm[0], tm[0]
        for i in range(1, 24):
            pa = self.par[i]
            Gr[i] = Gr[pa] @ Rm[i]
            Gt[i] = Gr[pa] @ tm[i] + Gt[pa]
        At = Gt - np.einsum('jab,jb->ja', Gr, J)                              # lbs.py:365-368
        # skinning: lbs.py:207-220
        Tr = np.einsum('vj,jab->vab', self.W, Gr)
        Tt = self.W @ At
        x = np.einsum('vab,vb->va', Tr, v_posed) + Tt
        kp = self.Ksel @ x + tau                                              # body_models_scale.py:393-403
        verts = x + tau
        out = dict(vertices=verts, joints=kp, body_pose=body_pose, full_pose=theta.reshape(72),
                   A_rot=Gr, A_trans=At, J=J, v_posed=v_posed, pose_feature=pf)
        if want_cache:
            out['_cache'] = (beta, theta, tau, s, J, R, rc, v_posed, Rm, tm, Gr, Gt, Tr, cache_vp)
        return out

    # -- SMPLifyLoss.forward: fitting.py:290-415 (no SDF term here; see oracle/sdf_np.py)
    def loss_terms(self, out, cams, gt_xy, w_conf, wts, use_vposer, pose_embedding=None,
                   prior=PRIOR_L2, fix_shape=False, beta=None, joints3d=None):
        dt = self.dtype
        cam_R, cam_t, cam_f, cam_c = (np.asarray(a, dt) for a in cams)
        kp = out['joints']
        rho2 = dt(wts['rho']) ** 2
        dw2 = dt(wts['data_weight']) ** 2
        p = np.einsum('vab,kb->vka', cam_R, kp) + cam_t[:, None, :]          # camera.py:106-110
        uv = cam_f[:, None, None] * p[..., :2] / p[..., 2:3] + cam_c[:, None, :]  # camera.py:112-116
        r = np.asarray(gt_xy, dt) - uv
        r2 = r * r
        gm = rho2 * r2 / (r2 + rho2)                                          # utils.py:435-438
        w2 = (np.asarray(w_conf, dt) ** 2)[..., None]
        L_data = (w2 * gm).sum() * dw2                                        # fitting.py:311-316
        r3 = None
        if joints3d is not None:                                              # fitting.py:319-324 (use_3d)
            gt3d, conf3d = joints3d
            r3 = np.asarray(gt3d, dt) - kp
            gm3 = rho2 * r3 * r3 / (r3 * r3 + rho2)
            L_data = L_data + ((np.asarray(conf3d, dt) ** 2)[:, None] * gm3).sum() * dw2
        wp = dt(wts['body_pose_weight'])
        bp = out['body_pose']
        gmm_sel = -1
        if use_vposer:
            z = np.asarray(pose_embedding, dt).reshape(32)
            L_pose = (z * z).sum() * wp ** 2                                  # fitting.py:327-329
            pose_dropped = False
        else:
            if prior == PRIOR_L2:
                P = (bp * bp).sum()                                           # prior.py:92-97
            else:
                means, prec, nllw = self.gmm
                dm = bp[None, :] - means                                      # prior.py:181-196
                quad = np.einsum('mi,mij,mj->m', dm, prec, dm)
                ll = 0.5 * quad - np.log(nllw)
                gmm_sel = int(np.argmin(ll))
                P = ll[gmm_sel]
            P = P * wp ** 2
            pose_dropped = bool(float(P) > 5e4)                               # fitting.py:334-335
            if pose_dropped:
                P = dt(0.0)
            L_pose = P + (bp * bp).sum() * (wp * 4) ** 2                      # fitting.py:336-337
        L_shape = dt(0.0)
        if not fix_shape:
            b = np.asarray(beta, dt)
            L_shape = (b * b).sum() * dt(wts['shape_weight']) ** 2           # fitting.py:339-342
        ang = out['full_pose'][3:66][ANGLE_IDX] * ANGLE_SGN
        L_angle = (np.exp(ang) ** 2).sum() * dt(wts['bending_prior_weight'])  # fitting.py:345-348
        angle_dropped = bool(float(L_angle) > 1e4 and not use_vposer)         # fitting.py:349-350
        if angle_dropped:
            L_angle = dt(0.0)
        total = L_data + L_pose + L_shape + L_angle
        return total, dict(L_data=L_data, L_pose=L_pose, L_shape=L_shape, L_angle=L_angle,
                           p=p, uv=uv, r=r, r3=r3, pose_dropped=pose_dropped,
                           angle_dropped=angle_dropped, gmm_sel=gmm_sel)

    def closure(self, x_flat, cams, gt_xy, w_conf, wts, use_vposer=False, prior=PRIOR_L2,
                fix_shape=False, g_verts_extra=None, joints3d=None, sdf=None):
        """loss, grad[D], out   for one problem.  ``g_verts_extra`` [6890,3] optionally adds
        an external dL/dvertices.  ``sdf`` = dict(faces, num_faces, grid_size) adds the interpenetration
        term of fitting.py:352-393 (oracle/sdf_term_np.py) with weight wts['coll_loss_weight']."""
        dt = self.dtype
        x_flat = np.asarray(x_flat, dt)
        p = dict(unpack(x_flat, use_vposer))
        p['use_vposer'] = use_vposer
        out = self.body(p)
        z = p.get('pose_embedding')
        total, aux = self.loss_terms(out, cams, gt_xy, w_conf, wts, use_vposer, z, prior,
                                     fix_shape, p['betas'], joints3d=joints3d)
        if sdf is not None and float(wts.get('coll_loss_weight', 0.0)) > 0:
            from oracle import sdf_term_np
            pen, g_sdf, sdf_aux = sdf_term_np.sdf_term(out['vertices'], sdf['faces'], wts['coll_loss_weight'],
                                                       sdf.get('num_faces', 1), sdf.get('grid_size', 128), dt)
            total = total + pen
            out['sdf'] = sdf_aux
            g_verts_extra = g_sdf if g_verts_extra is None else g_verts_extra + g_sdf
        grad = self._backward(out, aux, cams, w_conf, wts, use_vposer, z, prior, fix_shape,
                              g_verts_extra, joints3d=joints3d)
        return total, grad, out

    # -- reverse mode (hand-derived; replaces autograd of fitting.py:190-192)
    def _backward(self, out, aux, cams, w_conf, wts, use_vposer, z, prior, fix_shape,
                  g_verts_extra=None, joints3d=None):
        dt = self.dtype
        beta, theta, tau, s, J, R, rc, v_posed, Rm, tm, Gr, Gt, Tr, cache_vp = out['_cache']
        cam_R, cam_t, cam_f, cam_c = (np.asarray(a, dt) for a in cams)
        rho2 = dt(wts['rho']) ** 2
        dw2 = dt(wts['data_weight']) ** 2
        wp = dt(wts['body_pose_weight'])
        r, p = aux['r'], aux['p']
        w2 = (np.asarray(w_conf, dt) ** 2)[..., None]
        dgm = 2.0 * r * rho2 * rho2 / (r * r + rho2) ** 2
        g_uv = -w2 * dw2 * dgm                                                # [V,17,2]
        f = cam_f[:, None]
        pz = p[..., 2]
        g_p = np.stack([f * g_uv[..., 0] / pz, f * g_uv[..., 1] / pz,
                        -f * (g_uv[..., 0] * p[..., 0] + g_uv[..., 1] * p[..., 1]) / (pz * pz)],
                       axis=-1)
        g_kp = np.einsum('vab,vka->kb', cam_R, g_p)                           # [17,3]
        if joints3d is not None:
            r3 = aux['r3']
            c2 = (np.asarray(joints3d[1], dt) ** 2)[:, None]
            g_kp = g_kp - c2 * dw2 * 2.0 * r3 * rho2 * rho2 / (r3 * r3 + rho2) ** 2
        g_tau = g_kp.sum(0)
        gx = self.Ksel.T @ g_kp                                               # [Nv,3]
        if g_verts_extra is not None:
            gx = gx + g_verts_extra
            g_tau = g_tau + g_verts_extra.sum(0)
        # skinning
        g_vposed = np.einsum('vab,va->vb', Tr, gx)
        g_Ar = np.einsum('vj,va,vb->jab', self.W, gx, v_posed)
        g_At = self.W.T @ gx
        g_Gt = g_At.copy()
        g_Gr = g_Ar - np.einsum('ja,jb->jab', g_At, J)
        g_J = -np.einsum('jab,ja->jb', Gr, g_At)
        g_Rm = np.zeros_like(Gr)
        g_tm = np.zeros_like(Gt)
        for i in range(23, 0, -1):
            pa = self.par[i]
            g_Rm[i] = Gr[pa].T @ g_Gr[i]
            g_tm[i] = Gr[pa].T @ g_Gt[i]
            g_Gr[pa] += g_Gr[i] @ Rm[i].T + np.outer(g_Gt[i], tm[i])
            g_Gt[pa] += g_Gt[i]
        g_Rm[0], g_tm[0] = g_Gr[0], g_Gt[0]
        g_J += g_tm
        for i in range(1, 24):
            g_J[self.par[i]] -= g_tm[i]
        g_s = (g_Rm[0] * R[0]).sum()
        g_R = g_Rm.copy()
        g_R[0] = s * g_Rm[0]
        g_pf = self.PD @ g_vposed.reshape(-1)
        g_R[1:] += g_pf.reshape(23, 3, 3)
        g_vshaped = g_vposed + self.JR.T @ g_J
        g_beta = np.einsum('vkl,vk->l', self.S, g_vshaped)
        g_theta = np.zeros((24, 3), dt)
        for i in range(24):
            g_theta[i] = rodrigues_bwd(g_R[i], theta[i], rc[i])
        g_theta = g_theta.reshape(72)
        # priors
        bp = out['body_pose']
        g_z = None
        if use_vposer:
            g_z = 2.0 * np.asarray(z, dt) * wp ** 2
        else:
            if not aux['pose_dropped']:
                if prior == PRIOR_L2:
                    g_theta[3:] += 2.0 * bp * wp ** 2
                else:
                    means, prec, nllw = self.gmm
                    m = aux['gmm_sel']
                    dm = bp - means[m]
                    g_theta[3:] += 0.5 * (prec[m] @ dm + prec[m].T @ dm) * wp ** 2
            g_theta[3:] += 2.0 * bp * (4 * wp) ** 2
        if not fix_shape:
            g_beta = g_beta + 2.0 * beta * dt(wts['shape_weight']) ** 2
        if not aux['angle_dropped']:
            idx = ANGLE_IDX + 3
            g_theta[idx] += 2.0 * np.exp(2.0 * theta.reshape(72)[idx] * ANGLE_SGN) * ANGLE_SGN \
                * dt(wts['bending_prior_weight'])
        if use_vposer:
            g_z = g_z + vposer_decode_bwd(g_theta[3:], self.vp, cache_vp)
            return np.concatenate([g_beta, g_theta[:3], g_tau, [g_s], g_z])
        return np.concatenate([g_beta, g_theta[:3], g_theta[3:], g_tau, [g_s]])
