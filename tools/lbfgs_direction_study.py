"""Developer study (CPU, build container): is the COMPACT form of the L-BFGS direction with an explicitly maintained
R^-1 (float32 storage) as robust as the two-loop recursion in float32 on the real objective?

Runs full 4-stage fits of synthetic frames with the PyTorch-CPU port of the closure (oracle/closure_torch.py) under the
NumPy restatement of the reference optimiser (oracle/lbfgs_np.py), once with the reference's two-loop recursion and once
with the direction replaced by the compact (Byrd-Nocedal-Schnabel) form the device uses:

    p = S q ; w = R^-1 p ; t = Y^T w - q ; z = D w + gamma Y t ; a = R^-T z ; d = S^T a - gamma t

R^-1 maintained one bordered column per accepted pair, dropped row / column on eviction.  Reports closures, final
losses and the first-stage trajectory distance between the two.  Test infrastructure only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mvsmplfitting_amd import synthetic as syn  # noqa: E402
from mvsmplfitting_amd.engine import stage_weights  # noqa: E402
from oracle import closure_np as cn  # noqa: E402
from oracle import closure_torch as ct  # noqa: E402
from oracle import lbfgs_np as ln  # noqa: E402


class CompactOracle(ln.LbfgsOracle):
    """Same optimiser, direction in compact form.  acc = accumulation dtype of the small m x m algebra."""

    def __init__(self, *a, acc=np.float32, store=np.float32, **k):
        super().__init__(*a, **k)
        self.acc, self.store = acc, store
        self.Minv = np.zeros((0, 0), store)

    def step(self):
        # identical to LbfgsOracle.step except for the direction block
        orig_loss, g = self._eval(self.x)
        loss = orig_loss
        cur_evals = 1
        if np.abs(g).max() <= self.tol_grad:
            return orig_loss
        d, t, H = self.d, self.t, self.H
        n = 0
        f32 = self.x.dtype
        while n < self.max_iter:
            n += 1
            self.n_iter += 1
            if self.n_iter == 1:
                d = -g
                self.dirs, self.stps, self.ro = [], [], []
                self.Minv = np.zeros((0, 0), self.store)
                H = 1.0
            else:
                y = g - self.prev_g
                s = d * t
                ys = float(y @ s)
                if ys > 1e-10:
                    if len(self.dirs) == self.history:
                        self.dirs.pop(0); self.stps.pop(0); self.ro.pop(0)
                        self.Minv = self.Minv[1:, 1:]
                    k0 = len(self.dirs)
                    if k0:
                        Sm = np.stack(self.stps)
                        u = (Sm @ y).astype(f32)                                     # s_i . y_new
                        c = -(np.triu(self.Minv).astype(self.acc) @ u.astype(self.acc)) * self.acc(1.0 / ys)
                    else:
                        c = np.zeros(0, self.acc)
                    Mn = np.zeros((k0 + 1, k0 + 1), self.store)
                    Mn[:k0, :k0] = self.Minv
                    Mn[:k0, k0] = c.astype(self.store)
                    Mn[k0, k0] = self.store(1.0 / ys)
                    self.Minv = Mn
                    self.dirs.append(y); self.stps.append(s); self.ro.append(1.0 / ys)
                    H = ys / float(y @ y)
                k = len(self.dirs)
                q = -g
                if k:
                    Sm, Ym = np.stack(self.stps), np.stack(self.dirs)
                    Mi = np.triu(self.Minv).astype(self.acc)
                    gam = f32.type(H)
                    p = (Sm @ q).astype(f32)
                    w = (Mi @ p.astype(self.acc)).astype(f32)
                    tt = (Ym.T @ w).astype(f32) - q
                    v = (Ym @ tt).astype(f32)
                    z = (w.astype(self.acc) / np.asarray(self.ro, self.acc) + self.acc(H) * v.astype(self.acc))
                    a = (Mi.T @ z).astype(f32)
                    d = (Sm.T @ a).astype(f32) - gam * tt
                else:
                    d = q * f32.type(H)
            self.prev_g = g.copy()
            self.prev_loss = loss
            if self.n_iter == 1:
                t = min(1.0, 1.0 / float(np.abs(g).sum())) * self.lr
            else:
                t = self.lr
            gtd = float(g @ d)
            if gtd > -self.tol_change:
                self.exits.append(('gtd', self.n_iter))
                break
            loss, g, t, ls_evals = self._strong_wolfe(t, d, loss, g, gtd)
            self.x = self.x + t * d
            cur_evals += ls_evals
            if n == self.max_iter or cur_evals >= self.max_eval:
                break
            if np.abs(g).max() <= self.tol_grad or np.abs(d * t).max() <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        self.d, self.t, self.H = d, t, H
        return orig_loss


def fit(tc, x0, stages, mk, use_vp=False):
    lay, D = cn.param_layout(use_vp)
    segs = [lay[k] for k in lay]
    x = np.array(x0, np.float64)
    ncl, final, traces, gtd_exits = 0, None, [], 0
    for wts in stages:
        opt = mk(x, lambda xx: tc.evaluate(xx, wts, use_vp))
        final, _ = ln.run_fitting(opt, maxiters=30, segments=segs)
        x = opt.x.astype(np.float64)
        ncl += opt.func_evals
        traces.append(opt.trace)
        gtd_exits += sum(1 for e in opt.exits if e[0] == 'gtd')
    return x, final, ncl, traces, gtd_exits


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    use_vp = len(sys.argv) > 2 and sys.argv[2] == 'vposer'
    torch.set_num_threads(1)
    model = syn.make_body_model(0, skin_topk=4)
    cams = syn.make_camera_ring(8)
    vpw = syn.make_vposer_decoder() if use_vp else None
    orc = cn.ClosureOracle(model, np.float64)
    fr = syn.make_frames(nfr, seed0=1000)
    kp = np.stack([orc.body(dict({k: fr[k][b] for k in fr}, use_vposer=False), want_cache=False)['joints'] for b in range(nfr)])
    gt, conf = syn.make_observations(kp, cams, seed=1007)
    stages = stage_weights(1536.0)
    lay, D = cn.param_layout(use_vp)
    x0 = np.zeros(D); x0[lay['scale'][0]] = 1.0
    variants = [('two-loop f32', lambda x, f: ln.LbfgsOracle(x, f, dtype=np.float32)),
                ('compact store f32 acc f32', lambda x, f: CompactOracle(x, f, dtype=np.float32, acc=np.float32, store=np.float32)),
                ('compact store f32 acc f64', lambda x, f: CompactOracle(x, f, dtype=np.float32, acc=np.float64, store=np.float32)),
                ('compact store f64 acc f64', lambda x, f: CompactOracle(x, f, dtype=np.float32, acc=np.float64, store=np.float64))]
    for b in range(nfr):
        tc = ct.TorchClosure(model, cams, gt[b], conf[b], vposer=vpw)
        base = None
        for name, mk in variants:
            t0 = time.time()
            x, final, ncl, traces, ge = fit(tc, x0, stages, mk, use_vp)
            if base is None:
                base = traces
                dist = ''
            else:
                n = min(len(base[0]), len(traces[0]), 35)
                dd = [np.abs(base[0][k][0] - traces[0][k][0]).max() for k in range(n)]
                dist = ' | stage-0 x distance to two-loop at closure 5/15/25/34: ' + ' '.join('%.1e' % dd[min(k, n - 1)] for k in (5, 15, 25, 34))
            print('frame %d %-28s final %.4f closures %4d gtd-exits %d (%.1fs)%s' % (b, name, final, ncl, ge, time.time() - t0, dist), flush=True)


if __name__ == '__main__':
    main()
