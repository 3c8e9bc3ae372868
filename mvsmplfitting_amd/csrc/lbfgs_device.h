// Device-resident L-BFGS with strong-Wolfe line search + the run_fitting outer loop, as a
// resumable per-problem state machine executed by ONE wave64.
//
// What it restates (reference file:line):
//   LBFGS.step                      code/optimizers/lbfgs_ls.py:256-445
//   _strong_Wolfe/_cubic_interpolate code/optimizers/lbfgs_ls.py:39-167, 11-36
//   FittingMonitor.run_fitting      code/utils/fitting.py:99-142  (+ rel_change utils.py:348-349)
//   stage loop (fresh optimiser)    code/utils/non_linear_solver.py:156-211
//
// The reference calls closure() from inside nested Python loops; here every closure call is a
// yield point: advance() consumes (loss, grad) of the last trial point, runs the optimiser
// logic up to the next closure call, emits the next trial point and returns.  Vectors are
// lane-distributed (element i lives in lane i%64, register i/64); all scalars are wave-uniform
// (xor-butterfly reductions give bit-identical values in every lane), so there is no LDS and
// no barrier in here.  VT = vector storage type (float in production, double in the KAT);
// scalar logic and dot-product accumulation are always double.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvfit {

constexpr int LB_NPL = 2;          // elements per lane -> D <= 128
constexpr int LB_HIST = 100;

enum LbPhase : int { PH_STEP_START = 0, PH_LS_FIRST = 1, PH_LS_BRACKET = 2, PH_LS_ZOOM = 3 };

struct LbOpts {
    double lr, tol_grad, tol_change, ftol, gtol;
    int max_iter, max_eval, history, maxiters, num_stages;
    int nseg;            // parameter tensors taking part in the gtol test
    int seg_lo[8], seg_hi[8];
};

// Per-problem scalar state (lives in global memory between launches).
struct LbState {
    int phase, status, stage, outer_n;
    int n_iter, n, cur_evals;
    int hist_len, hist_head;
    int ls_it, ls_evals, ls_done, low, high, insuf, nbr;
    int has_outer_prev;
    int n_closure, n_lbfgs;
    int pad0;
    double loss, prev_loss, orig_loss, outer_prev;
    double t, H, gtd, f0, d_norm;
    double t_prev, f_prev, gtd_prev;
    double br[2], bf[2], bgtd[2];
    double stage_final[8];
};

template <typename VT>
struct LbVecs {            // lane-distributed working vectors
    VT x[LB_NPL], d[LB_NPL], g[LB_NPL], pg[LB_NPL], gprev[LB_NPL], bg0[LB_NPL], bg1[LB_NPL];
};

// History ring in global memory: y = dirs, s = stps, row stride LB_D_STRIDE.
constexpr int LB_D_STRIDE = 128;
template <typename VT>
struct LbHist {
    VT* dirs;      // [LB_HIST][128]
    VT* stps;      // [LB_HIST][128]
    double* ro;    // [LB_HIST]
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename VT>
__device__ __forceinline__ double vdot(const VT* a, const VT* b) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) s += (double)a[r] * (double)b[r];
    return wave_sum(s);
}
template <typename VT>
__device__ __forceinline__ double vmaxabs(const VT* a) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) s = fmax(s, fabs((double)a[r]));
    return wave_max(s);
}

// lbfgs_ls.py:11-36
__device__ __forceinline__ double lb_cubic(double x1, double f1, double g1, double x2, double f2,
                                           double g2, bool has_bounds, double lo, double hi) {
    if (!has_bounds) {
        if (x1 <= x2) { lo = x1; hi = x2; } else { lo = x2; hi = x1; }
    }
    double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
    double d2s = d1 * d1 - g1 * g2;
    if (d2s >= 0.0) {
        double d2 = sqrt(d2s);
        double mp;
        if (x1 <= x2) mp = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2));
        else          mp = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2));
        return fmin(fmax(mp, lo), hi);
    }
    return (lo + hi) / 2.0;
}

// Consume (f_new, gnew) of the last closure call, emit the next trial point into xt.
// Returns with S.status == 1 when all stages are finished (xt = final x).
// D <= 128; lane = threadIdx & 63.  mask[r] = 0 freezes an element (gradient forced to zero by
// the caller already; kept here only for the gtol segments).
template <typename VT>
__device__ void lbfgs_advance(LbState& S, LbVecs<VT>& V, const LbHist<VT>& Hh, const LbOpts& O,
                              double f_new, const VT* gnew, VT* xt, int lane, int D) {
    const double c1 = 1e-4, c2 = 0.9;
    const int max_ls = 25;
    double gtd_new = 0.0;
    int idx[LB_NPL];
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) idx[r] = lane + 64 * r;

    S.n_closure += 1;
    switch (S.phase) {
        case PH_LS_FIRST: goto L_ls_first;
        case PH_LS_BRACKET: goto L_ls_bracket;
        case PH_LS_ZOOM: goto L_ls_zoom;
        default: break;
    }

    // ---- lbfgs_ls.py:280-290 : closure at the start of step() ----
    S.orig_loss = f_new;
    S.loss = f_new;
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) V.g[r] = gnew[r];
    S.cur_evals = 1;
    if (vmaxabs(V.g) <= O.tol_grad) goto L_step_return;
    S.n = 0;

L_iter:
    S.n += 1;
    S.n_iter += 1;
    S.n_lbfgs += 1;
    if (S.n_iter == 1) {                                              // :312-317
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) V.d[r] = -V.g[r];
        S.hist_len = 0;
        S.hist_head = 0;
        S.H = 1.0;
    } else {                                                          // :318-358
        VT y[LB_NPL], s[LB_NPL];
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) {
            y[r] = V.g[r] - V.pg[r];
            s[r] = (VT)((double)V.d[r] * S.t);
        }
        double ys = vdot(y, s);
        if (ys > 1e-10) {
            if (S.hist_len == O.history) S.hist_head = (S.hist_head + 1) % LB_HIST;
            else S.hist_len += 1;
            int slot = (S.hist_head + S.hist_len - 1) % LB_HIST;
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) {
                Hh.dirs[slot * LB_D_STRIDE + idx[r]] = y[r];
                Hh.stps[slot * LB_D_STRIDE + idx[r]] = s[r];
            }
            if (lane == 0) Hh.ro[slot] = 1.0 / ys;
            S.H = ys / vdot(y, y);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // ro[] is written by lane 0, read by all
        }
        // two-loop recursion; al[i] kept lane-distributed (lane i%64, register i/64)
        double al0 = 0.0, al1 = 0.0;
        VT q[LB_NPL];
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) q[r] = -V.g[r];
        for (int i = S.hist_len - 1; i >= 0; --i) {
            int slot = (S.hist_head + i) % LB_HIST;
            VT sv[LB_NPL], yv[LB_NPL];
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) {
                sv[r] = Hh.stps[slot * LB_D_STRIDE + idx[r]];
                yv[r] = Hh.dirs[slot * LB_D_STRIDE + idx[r]];
            }
            double a = vdot(sv, q) * Hh.ro[slot];
            if (lane == (i & 63)) { if (i < 64) al0 = a; else al1 = a; }
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) q[r] = (VT)((double)q[r] - a * (double)yv[r]);
        }
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) q[r] = (VT)((double)q[r] * S.H);
        for (int i = 0; i < S.hist_len; ++i) {
            int slot = (S.hist_head + i) % LB_HIST;
            VT sv[LB_NPL], yv[LB_NPL];
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) {
                sv[r] = Hh.stps[slot * LB_D_STRIDE + idx[r]];
                yv[r] = Hh.dirs[slot * LB_D_STRIDE + idx[r]];
            }
            double be = vdot(yv, q) * Hh.ro[slot];
            double a = __shfl((i < 64) ? al0 : al1, i & 63, 64);
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) q[r] = (VT)((double)q[r] + (a - be) * (double)sv[r]);
        }
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) V.d[r] = q[r];
    }
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) V.pg[r] = V.g[r];                // :360-364
    S.prev_loss = S.loss;
    if (S.n_iter == 1) {                                              // :370-373
        double asum = 0.0;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) asum += fabs((double)V.g[r]);
        asum = wave_sum(asum);
        S.t = fmin(1.0, 1.0 / asum) * O.lr;
    } else {
        S.t = O.lr;
    }
    S.gtd = vdot(V.g, V.d);                                           // :376
    if (S.gtd > -O.tol_change) goto L_step_return;                    // :379-380
    // ---- _strong_Wolfe entry (:43-53) ----
    S.d_norm = vmaxabs(V.d);
    S.f0 = S.loss;
    S.ls_evals = 0;
    S.phase = PH_LS_FIRST;
    goto L_emit_trial;

L_ls_first:
    S.ls_evals = 1;
    gtd_new = vdot(gnew, V.d);
    S.t_prev = 0.0; S.f_prev = S.f0; S.gtd_prev = S.gtd;
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) V.gprev[r] = V.g[r];
    S.ls_done = 0;
    S.ls_it = 0;
    goto L_bracket_check;

L_ls_bracket:                                                         // :90-93
    S.ls_evals += 1;
    gtd_new = vdot(gnew, V.d);
    S.ls_it += 1;

L_bracket_check:                                                      // :54-93
    if (S.ls_it < max_ls) {
        bool two_point = false;
        if (f_new > (S.f0 + c1 * S.t * S.gtd) || (S.ls_it > 1 && f_new >= S.f_prev)) {
            two_point = true;                                         // :56-61
        } else if (fabs(gtd_new) <= -c2 * S.gtd) {                    // :63-69
            S.br[0] = S.t; S.bf[0] = f_new; S.bgtd[0] = gtd_new;
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) V.bg0[r] = gnew[r];
            S.nbr = 1;
            S.ls_done = 1;
            goto L_bracket_end;
        } else if (gtd_new >= 0.0) {                                  // :71-76
            two_point = true;
        }
        if (two_point) {
            S.br[0] = S.t_prev; S.br[1] = S.t;
            S.bf[0] = S.f_prev; S.bf[1] = f_new;
            S.bgtd[0] = S.gtd_prev; S.bgtd[1] = gtd_new;
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) { V.bg0[r] = V.gprev[r]; V.bg1[r] = gnew[r]; }
            S.nbr = 2;
            goto L_bracket_end;
        }
        {                                                             // :78-89
            double min_step = S.t + 0.01 * (S.t - S.t_prev);
            double max_step = S.t * 10.0;
            double tmp = S.t;
            S.t = lb_cubic(S.t_prev, S.f_prev, S.gtd_prev, S.t, f_new, gtd_new, true, min_step, max_step);
            S.t_prev = tmp; S.f_prev = f_new; S.gtd_prev = gtd_new;
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) V.gprev[r] = gnew[r];
        }
        S.phase = PH_LS_BRACKET;
        goto L_emit_trial;
    }

L_bracket_end:
    if (S.ls_it == max_ls) {                                          // :96-100
        S.br[0] = 0.0; S.br[1] = S.t;
        S.bf[0] = S.f0; S.bf[1] = f_new;
        S.bgtd[0] = S.gtd; S.bgtd[1] = gtd_new;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) { V.bg0[r] = V.g[r]; V.bg1[r] = gnew[r]; }
        S.nbr = 2;
    }
    S.insuf = 0;
    if (S.bf[0] <= S.bf[S.nbr - 1]) { S.low = 0; S.high = 1; } else { S.low = 1; S.high = 0; }

L_zoom_check:                                                         // :108-130
    if (!S.ls_done && S.ls_it < O.max_iter) {
        double tt = lb_cubic(S.br[0], S.bf[0], S.bgtd[0], S.br[1], S.bf[1], S.bgtd[1], false, 0, 0);
        double bmax = fmax(S.br[0], S.br[1]), bmin = fmin(S.br[0], S.br[1]);
        double eps = 0.1 * (bmax - bmin);
        if (fmin(bmax - tt, tt - bmin) < eps) {
            if (S.insuf || tt >= bmax || tt <= bmin) {
                if (fabs(tt - bmax) < fabs(tt - bmin)) tt = bmax - eps; else tt = bmin + eps;
                S.insuf = 0;
            } else {
                S.insuf = 1;
            }
        } else {
            S.insuf = 0;
        }
        S.t = tt;
        S.phase = PH_LS_ZOOM;
        goto L_emit_trial;
    }
    goto L_ls_return;

L_ls_zoom:                                                            // :130-161
    S.ls_evals += 1;
    gtd_new = vdot(gnew, V.d);
    S.ls_it += 1;
    if (f_new > (S.f0 + c1 * S.t * S.gtd) || f_new >= S.bf[S.low]) {
        int h = S.high;
        S.br[h] = S.t; S.bf[h] = f_new; S.bgtd[h] = gtd_new;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) { if (h == 0) V.bg0[r] = gnew[r]; else V.bg1[r] = gnew[r]; }
        if (S.bf[0] <= S.bf[1]) { S.low = 0; S.high = 1; } else { S.low = 1; S.high = 0; }
    } else {
        if (fabs(gtd_new) <= -c2 * S.gtd) {
            S.ls_done = 1;
        } else if (gtd_new * (S.br[S.high] - S.br[S.low]) >= 0.0) {
            int h = S.high, l = S.low;
            S.br[h] = S.br[l]; S.bf[h] = S.bf[l]; S.bgtd[h] = S.bgtd[l];
#pragma unroll
            for (int r = 0; r < LB_NPL; ++r) { if (h == 0) V.bg0[r] = V.bg1[r]; else V.bg1[r] = V.bg0[r]; }
        }
        int l = S.low;
        S.br[l] = S.t; S.bf[l] = f_new; S.bgtd[l] = gtd_new;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) { if (l == 0) V.bg0[r] = gnew[r]; else V.bg1[r] = gnew[r]; }
    }
    if (fabs(S.br[1] - S.br[0]) * S.d_norm < O.tol_change) goto L_ls_return;
    goto L_zoom_check;

L_ls_return:                                                          // :163-167, :393-399
    {
        int l = S.low;
        S.loss = S.bf[l];
        S.t = S.br[l];
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) {
            V.g[r] = (l == 0) ? V.bg0[r] : V.bg1[r];
            V.x[r] = (VT)((double)V.x[r] + S.t * (double)V.d[r]);
        }
        S.cur_evals += S.ls_evals;
    }
    if (S.n == O.max_iter) goto L_step_return;                        // :419-434
    if (S.cur_evals >= O.max_eval) goto L_step_return;
    if (vmaxabs(V.g) <= O.tol_grad) goto L_step_return;
    {
        double m = 0.0;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) m = fmax(m, fabs((double)V.d[r] * S.t));
        if (wave_max(m) <= O.tol_change) goto L_step_return;
    }
    if (fabs(S.loss - S.prev_loss) < O.tol_change) goto L_step_return;
    goto L_iter;

L_step_return:
    {   // ---- run_fitting, fitting.py:100-142, with loss = orig_loss (lbfgs_ls.py:445) ----
        double loss_out = S.orig_loss;
        bool stop = false;
        if (isnan(loss_out) || isinf(loss_out)) {
            stop = true;
        } else {
            if (S.outer_n > 0 && S.has_outer_prev && O.ftol > 0.0) {
                double den = fmax(fmax(fabs(S.outer_prev), fabs(loss_out)), 1.0);
                if ((S.outer_prev - loss_out) / den <= O.ftol) stop = true;
            }
            if (!stop) {
                // all(|max(grad_tensor)| < gtol): grad of the LAST closure call (= gnew)
                bool all_small = true;
                for (int sgi = 0; sgi < O.nseg; ++sgi) {
                    double m = -INFINITY;
#pragma unroll
                    for (int r = 0; r < LB_NPL; ++r)
                        if (idx[r] >= O.seg_lo[sgi] && idx[r] < O.seg_hi[sgi]) m = fmax(m, (double)gnew[r]);
                    m = wave_max(m);
                    if (!(fabs(m) < O.gtol)) all_small = false;
                }
                if (all_small) stop = true;
            }
            if (!stop) { S.outer_prev = loss_out; S.has_outer_prev = 1; }
        }
        S.outer_n += 1;
        if (stop || S.outer_n >= O.maxiters) {
            S.stage_final[S.stage] = S.has_outer_prev ? S.outer_prev : (double)NAN;
            S.stage += 1;
            S.outer_n = 0;
            S.has_outer_prev = 0;
            S.n_iter = 0;                 // fresh optimiser object (non_linear_solver.py:172)
            S.hist_len = 0;
            S.hist_head = 0;
            if (S.stage >= O.num_stages) S.status = 1;
        }
        S.phase = PH_STEP_START;
#pragma unroll
        for (int r = 0; r < LB_NPL; ++r) xt[r] = V.x[r];
        return;
    }

L_emit_trial:                                                         // _directional_evaluate :249-254
#pragma unroll
    for (int r = 0; r < LB_NPL; ++r) xt[r] = (VT)((double)V.x[r] + S.t * (double)V.d[r]);
    (void)D;
    return;
}

}  // namespace mvfit
