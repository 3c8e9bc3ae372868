"""TESTS ONLY - a stand-in for mvsmplfitting_amd.engine.MvFit that runs on the CPU by calling the ORACLE.

It exists so that the host-side mirror (mvsmplfitting_amd/fitting.py) can be executed by the reference's own,
unmodified caller (code/utils/non_linear_solver.py) in the build container, where there is no GPU: every call the
mirror makes on the engine is recorded, and the numbers come from oracle/closure_np.py + oracle/lbfgs_np.py in
float64.  The product never imports this file; on a GPU box the same mirror talks to libmvfit."""
import numpy as np
import torch

from mvsmplfitting_amd import _lib
from oracle import closure_np as cn
from oracle import lbfgs_np as ln

D = _lib.D


def _flat_layout(flags):
    """flat oracle vector <-> the C ABI's [118] layout (include/mvfit.h)."""
    vp = bool(flags & _lib.F_VPOSER)
    lay, n = cn.param_layout(vp)
    sl118 = dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86),
                 pose_embedding=(86, 118))
    return vp, lay, n, sl118


class StubMvFit:
    calls = []          # class-level log: (method, detail)

    def __init__(self, model, vposer=None, gmm=None, device=0):
        self.device = torch.device('cpu')
        self.dtype = torch.float64
        self.model = model
        self.orc = cn.ClosureOracle(model, np.float64, vposer=vposer, gmm=gmm)
        self.has_vposer, self.has_gmm = vposer is not None, gmm is not None
        self.B = self.V = 0
        self.j3 = None
        self.sdf_cfg = None
        StubMvFit.calls.append(('create', dict(vposer=vposer is not None, gmm=gmm is not None)))

    def set_problems(self, cams, gt_xy, w_conf):
        self.cams = tuple(np.asarray(a, np.float64) for a in cams)
        self.gt = np.asarray(gt_xy, np.float64)
        self.wc = np.asarray(w_conf, np.float64)
        self.B, self.V = self.gt.shape[0], self.gt.shape[1]
        StubMvFit.calls.append(('set_problems', (self.B, self.V)))

    def set_joints3d(self, gt3d, conf3d):
        self.j3 = (np.asarray(gt3d, np.float64), np.asarray(conf3d, np.float64))
        StubMvFit.calls.append(('set_joints3d', None))

    def set_sdf(self, faces, num_faces=1, grid_size=128):
        self.sdf_cfg = None if faces is None else dict(faces=np.asarray(faces), num_faces=num_faces, grid_size=grid_size)
        StubMvFit.calls.append(('set_sdf', faces is not None))

    def _eval_one(self, b, x118, w):
        flags = int(w.get('flags', 0))
        vp, lay, n, sl = _flat_layout(flags)
        xf = np.zeros(n)
        for name, (a, e) in lay.items():
            xf[a:e] = x118[sl[name][0]:sl[name][1]]
        j3 = None
        if flags & _lib.F_USE_3D:
            j3 = (self.j3[0][b], self.j3[1][b])
        L, g, _ = self.orc.closure(xf, self.cams, self.gt[b], self.wc[b], w, use_vposer=vp,
                                   prior=cn.PRIOR_GMM if flags & _lib.F_PRIOR_GMM else cn.PRIOR_L2,
                                   fix_shape=bool(flags & _lib.F_FIX_SHAPE), joints3d=j3, sdf=self.sdf_cfg)
        g118 = np.zeros(D)
        for name, (a, e) in lay.items():
            g118[sl[name][0]:sl[name][1]] = g[a:e]
        if flags & _lib.F_FIX_SHAPE:
            g118[0:10] = 0
        if flags & _lib.F_FIX_SCALE:
            g118[85] = 0
        return float(L), g118

    def closure(self, params, weights, want_grad=True, want_verts=False, want_joints=False):
        x = np.asarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, np.float64)
        StubMvFit.calls.append(('closure', dict(weights)))
        res = [self._eval_one(b, x[b], weights) for b in range(self.B)]
        out = dict(loss=torch.tensor([r[0] for r in res], dtype=torch.float64))
        if want_grad:
            out['grad'] = torch.tensor(np.stack([r[1] for r in res]))
        return out

    def fit(self, params, stages, lr=1.0, max_iter=30, history=100, tolerance_grad=1e-5, tolerance_change=1e-9,
            maxiters=30, ftol=1e-9, gtol=1e-9, max_rounds=0):
        """The device-resident staged fit, restated with the oracle optimiser (fresh state per stage, the same
        optimised-entry selection as include/mvfit.h documents)."""
        x = np.asarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, np.float64).copy()
        StubMvFit.calls.append(('fit', dict(n_stages=len(stages), maxiters=maxiters, max_iter=max_iter)))
        final = np.zeros(self.B)
        ncl = np.zeros(self.B, np.int32)
        nit = np.zeros(self.B, np.int32)
        for b in range(self.B):
            for w in stages:
                flags = int(w.get('flags', 0))
                vp, lay, n, sl = _flat_layout(flags)
                names = [k for k in lay if not (k == 'betas' and flags & _lib.F_FIX_SHAPE)
                         and not (k == 'scale' and flags & _lib.F_FIX_SCALE)]
                idx = np.concatenate([np.arange(*sl[k]) for k in names])
                segs, off = [], 0
                for k in names:
                    m = sl[k][1] - sl[k][0]
                    segs.append((off, off + m))
                    off += m

                def fn(z, b=b, w=w, idx=idx):
                    xx = x[b].copy()
                    xx[idx] = z
                    L, g = self._eval_one(b, xx, w)
                    return L, g[idx]
                opt = ln.LbfgsOracle(x[b, idx], fn, lr=lr, max_iter=max_iter, history=history,
                                     tol_grad=tolerance_grad, tol_change=tolerance_change)
                prev, losses = ln.run_fitting(opt, maxiters=maxiters, ftol=ftol, gtol=gtol, segments=segs)
                x[b, idx] = opt.x
                final[b] = np.nan if prev is None else prev
                ncl[b] += opt.func_evals
                nit[b] += opt.n_iter
        return torch.tensor(x), dict(final_loss=torch.tensor(final), n_closure=torch.tensor(ncl), n_iter=torch.tensor(nit))

    # ---- the rows either side of the path, answered by the oracle restatements (batch driver host-logic tests) ----
    def _params(self, x118, flags):
        x = np.asarray(x118.detach().cpu().numpy() if isinstance(x118, torch.Tensor) else x118, np.float64)
        p = dict(betas=x[0:10], global_orient=x[10:13], body_pose=x[13:82], transl=x[82:85], scale=x[85])
        if flags & _lib.F_VPOSER:
            p['pose_embedding'] = x[86:118]
        return p

    def vertices(self, params, flags=0):
        x = np.asarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, np.float64)
        outs = [self.orc.body(self._params(r, int(flags)), want_cache=False) for r in x]
        return (torch.tensor(np.stack([o['vertices'] for o in outs]), dtype=torch.float32),
                torch.tensor(np.stack([o['joints'] for o in outs]), dtype=torch.float32))

    def full_pose(self, params, flags=0):
        x = np.asarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, np.float64)
        return torch.tensor(np.stack([self.orc.body(self._params(r, int(flags)), want_cache=False)['full_pose'] for r in x]),
                            dtype=torch.float32)

    def triangulate(self, keypoints, intris, extris):
        from oracle import triangulate_np as tn
        kp = np.asarray(keypoints)
        return torch.tensor(np.stack([tn.recompute3d(extris, intris, kp[b]) for b in range(kp.shape[0])]))

    def depth_guess(self, rest_joints, extri, intri, keypoints):
        from oracle import init_guess_np as ign
        kp = np.asarray(keypoints.cpu().numpy() if isinstance(keypoints, torch.Tensor) else keypoints, np.float32)
        rest = np.asarray(rest_joints.cpu().numpy() if isinstance(rest_joints, torch.Tensor) else rest_joints, np.float64)
        return torch.as_tensor(np.stack([ign.single_view_joints3d(rest, extri, intri, kp[b]) for b in range(kp.shape[0])]),
                               dtype=torch.float64)

    def umeyama(self, src, dst, estimate_scale=True):
        from oracle import umeyama_np as un
        src = np.asarray(src.cpu().numpy() if isinstance(src, torch.Tensor) else src, np.float64)
        dst = np.asarray(dst.cpu().numpy() if isinstance(dst, torch.Tensor) else dst, np.float64)
        rot, tr, sc = [], [], []
        for b in range(dst.shape[0]):
            r, t, s_, _ = un.umeyama(src, dst[b], estimate_scale)
            rot.append(r); tr.append(t); sc.append(s_)
        rot = np.stack(rot)
        return dict(rot=torch.tensor(rot), rvec=torch.tensor(np.stack([un.rotvec(r) for r in rot])),
                    trans=torch.tensor(np.stack(tr)), scale=torch.tensor(np.asarray(sc, np.float64)))

    def close(self):
        pass

