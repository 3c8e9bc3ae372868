"""TEST INFRASTRUCTURE - NumPy restatement of the reference optimiser loop.

Restates, for a generic ``evalfn(x) -> (loss, grad)``:
  * ``LBFGS.step`` with strong-Wolfe line search  (reference code/optimizers/lbfgs_ls.py:256-445)
  * ``_strong_Wolfe`` / ``_cubic_interpolate``     (lbfgs_ls.py:39-167, 11-36)
  * ``FittingMonitor.run_fitting``                 (reference code/utils/fitting.py:71-142)
with the production settings of ``create_optimizer`` (optim_factory.py:50-52:
lr, max_iter=maxiters, max_eval=max_iter*5//4, tolerance_grad=1e-5,
tolerance_change=1e-9, history_size=100).

Pinned against the reference optimiser itself on analytic objectives
(tests/test_lbfgs_oracle.py; golden trajectories tests/golden/lbfgs_kat.npz).
The checker for the device L-BFGS; never imported by the shipped package.
"""
from __future__ import annotations

import math

import numpy as np


def cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    """lbfgs_ls.py:11-36"""
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2)
    d2s = d1 * d1 - g1 * g2
    if d2s >= 0:
        d2 = math.sqrt(d2s)
        if x1 <= x2:
            mp = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
        else:
            mp = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2))
        return min(max(mp, lo), hi)
    return (lo + hi) / 2.0


class LbfgsOracle:
    """State persists across ``step`` calls exactly like ``optimizer.state`` does."""

    def __init__(self, x0, evalfn, lr=1.0, max_iter=30, history=100, tol_grad=1e-5,
                 tol_change=1e-9, dtype=np.float64):
        self.x = np.array(x0, dtype)
        self.evalfn = evalfn
        self.lr, self.max_iter, self.history = lr, max_iter, history
        self.max_eval = max_iter * 5 // 4
        self.tol_grad, self.tol_change = tol_grad, tol_change
        self.n_iter = 0
        self.func_evals = 0
        self.d = self.t = self.H = self.prev_g = self.prev_loss = None
        self.dirs, self.stps, self.ro = [], [], []
        self.last_grad = None          # .grad left by the last closure call (gtol test reads it)
        self.trace = []                # (x_trial, loss) of every closure call
        self.exits = []                # why each step() ended: 'grad0', 'gtd' (:379-380, with n_iter), 'ls' (:419-434)

    def _eval(self, x):
        f, g = self.evalfn(x)
        f = float(f)
        self.last_grad = np.array(g, self.x.dtype)
        self.func_evals += 1
        self.trace.append((x.copy(), f))
        return f, self.last_grad.copy()

    def _strong_wolfe(self, t, d, f, g, gtd, c1=1e-4, c2=0.9, max_ls=25):
        """lbfgs_ls.py:39-167; obj_func(x,t,d) evaluates at x + t d (lbfgs_ls.py:249-254)."""
        tol, max_iter = self.tol_change, self.max_iter
        x = self.x
        d_norm = np.abs(d).max()
        f_new, g_new = self._eval(x + t * d)
        evals = 1
        gtd_new = float(g_new @ d)
        t_prev, f_prev, g_prev, gtd_prev = 0.0, f, g.copy(), gtd
        done = False
        it = 0
        br = None
        while it < max_ls:
            if f_new > (f + c1 * t * gtd) or (it > 1 and f_new >= f_prev):
                br = [t_prev, t]; bf = [f_prev, f_new]; bg = [g_prev, g_new.copy()]
                bgtd = [gtd_prev, gtd_new]
                break
            if abs(gtd_new) <= -c2 * gtd:
                br = [t]; bf = [f_new]; bg = [g_new]; bgtd = [gtd_new]
                done = True
                break
            if gtd_new >= 0:
                br = [t_prev, t]; bf = [f_prev, f_new]; bg = [g_prev, g_new.copy()]
                bgtd = [gtd_prev, gtd_new]
                break
            min_step = t + 0.01 * (t - t_prev)
            max_step = t * 10
            tmp = t
            t = cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new,
                                  bounds=(min_step, max_step))
            t_prev, f_prev, g_prev, gtd_prev = tmp, f_new, g_new.copy(), gtd_new
            f_new, g_new = self._eval(x + t * d)
            evals += 1
            gtd_new = float(g_new @ d)
            it += 1
        if it == max_ls:
            br = [0.0, t]; bf = [f, f_new]; bg = [g, g_new]; bgtd = [gtd, gtd_new]
        insuf = False
        low, high = (0, 1) if bf[0] <= bf[-1] else (1, 0)
        while not done and it < max_iter:
            t = cubic_interpolate(br[0], bf[0], bgtd[0], br[1], bf[1], bgtd[1])
            eps = 0.1 * (max(br) - min(br))
            if min(max(br) - t, t - min(br)) < eps:
                if insuf or t >= max(br) or t <= min(br):
                    if abs(t - max(br)) < abs(t - min(br)):
                        t = max(br) - eps
                    else:
                        t = min(br) + eps
                    insuf = False
                else:
                    insuf = True
            else:
                insuf = False
            f_new, g_new = self._eval(x + t * d)
            evals += 1
            gtd_new = float(g_new @ d)
            it += 1
            if f_new > (f + c1 * t * gtd) or f_new >= bf[low]:
                br[high] = t; bf[high] = f_new; bg[high] = g_new.copy(); bgtd[high] = gtd_new
                low, high = (0, 1) if bf[0] <= bf[1] else (1, 0)
            else:
                if abs(gtd_new) <= -c2 * gtd:
                    done = True
                elif gtd_new * (br[high] - br[low]) >= 0:
                    br[high] = br[low]; bf[high] = bf[low]; bg[high] = bg[low]
                    bgtd[high] = bgtd[low]
                br[low] = t; bf[low] = f_new; bg[low] = g_new.copy(); bgtd[low] = gtd_new
            if abs(br[1] - br[0]) * d_norm < tol:
                break
        return bf[low], bg[low], br[low], evals

    def step(self):
        """lbfgs_ls.py:256-445.  Returns the loss at the START of the step (orig_loss)."""
        orig_loss, g = self._eval(self.x)
        loss = orig_loss
        cur_evals = 1
        if np.abs(g).max() <= self.tol_grad:
            self.exits.append(('grad0', self.n_iter))
            return orig_loss
        d, t, H = self.d, self.t, self.H
        n = 0
        while n < self.max_iter:
            n += 1
            self.n_iter += 1
            if self.n_iter == 1:
                d = -g
                self.dirs, self.stps, self.ro = [], [], []
                H = 1.0
            else:
                y = g - self.prev_g
                s = d * t
                ys = float(y @ s)
                if ys > 1e-10:
                    if len(self.dirs) == self.history:
                        self.dirs.pop(0); self.stps.pop(0); self.ro.pop(0)
                    self.dirs.append(y); self.stps.append(s); self.ro.append(1.0 / ys)
                    H = ys / float(y @ y)
                k = len(self.dirs)
                al = [0.0] * k
                q = -g
                for i in range(k - 1, -1, -1):
                    al[i] = float(self.stps[i] @ q) * self.ro[i]
                    q = q - al[i] * self.dirs[i]
                r = q * H
                for i in range(k):
                    be = float(self.dirs[i] @ r) * self.ro[i]
                    r = r + (al[i] - be) * self.stps[i]
                d = r
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
            if n == self.max_iter:
                break
            if cur_evals >= self.max_eval:
                break
            if np.abs(g).max() <= self.tol_grad:
                break
            if np.abs(d * t).max() <= self.tol_change:
                break
            if abs(loss - self.prev_loss) < self.tol_change:
                break
        self.d, self.t, self.H = d, t, H
        return orig_loss


def run_fitting(opt: LbfgsOracle, maxiters=30, ftol=1e-9, gtol=1e-9, segments=None):
    """fitting.py:99-142.  ``segments``: list of (start, stop) of each parameter tensor in the
    flat vector - the gtol test is per tensor on abs(max(grad)) (fitting.py:115-116)."""
    prev = None
    losses = []
    D = opt.x.shape[0]
    segments = segments or [(0, D)]
    for n in range(maxiters):
        loss = opt.step()
        losses.append(loss)
        if math.isnan(loss) or math.isinf(loss):
            break
        if n > 0 and prev is not None and ftol > 0:
            rel = (prev - loss) / max(abs(prev), abs(loss), 1.0)          # utils.py:348-349
            if rel <= ftol:
                break
        if all(abs(opt.last_grad[a:b].max()) < gtol for a, b in segments):
            break
        prev = loss
    return prev, losses


# ------------------------------------------------------------ analytic KAT objectives
def kat_objective(kind: str, D: int, seed: int = 0):
    """Deterministic test objectives shared with the device KAT kernel
    (mvsmplfitting_amd/csrc/lbfgs_kat.hip implements the same formulas).

    'quad'  : 0.5 * sum c_i (x_i - m_i)^2,   c_i = 1 + 99 * i/(D-1), m_i = sin(i)
    'rosen' : chained Rosenbrock  sum_{i<D-1} 100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2
    'gmof'  : sum_i 1e4 * r_i^2/(r_i^2 + 1e4) with r_i = 50*(x_i - m_i) + 20 sin(3 x_{(i+1)%D})
              plus 0.5*sum x_i^2  (non-convex, GMoF-like)
    Returns (fn(x)->(f,g), x0).
    """
    i = np.arange(D, dtype=np.float64)
    m = np.sin(i)
    if kind == 'quad':
        c = 1.0 + 99.0 * i / (D - 1)

        def fn(x):
            r = x - m
            return 0.5 * float((c * r * r).sum()), c * r
        return fn, np.zeros(D)
    if kind == 'rosen':
        def fn(x):
            a = x[1:] - x[:-1] ** 2
            b = 1.0 - x[:-1]
            f = float((100.0 * a * a + b * b).sum())
            g = np.zeros(D)
            g[:-1] += -400.0 * a * x[:-1] - 2.0 * b
            g[1:] += 200.0 * a
            return f, g
        return fn, np.full(D, -1.2) * np.cos(0.1 * i)
    if kind == 'gmof':
        rho2 = 1e4

        def fn(x):
            xn = np.roll(x, -1)
            r = 50.0 * (x - m) + 20.0 * np.sin(3.0 * xn)
            r2 = r * r
            f = float((rho2 * r2 / (r2 + rho2)).sum() + 0.5 * (x * x).sum())
            dr = 2.0 * r * rho2 * rho2 / (r2 + rho2) ** 2
            g = 50.0 * dr + np.roll(dr * 60.0 * np.cos(3.0 * xn), 1) + x
            return f, g
        return fn, 0.3 * np.cos(0.7 * i)
    raise ValueError(kind)
