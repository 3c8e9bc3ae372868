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
// logic up to the next closure call, emits the next trial point and returns.
//
// Lane layout: the flat parameter vector (reference order of final_params, D <= 96) is spread
// over the wave, LB_EPL = 2 consecutive elements per lane (lanes 48.. hold zeros); every dot
// product is a 6-step butterfly (4 DPP + v_permlane16/32_swap, wave_ops.h) whose result is
// bit-identical in all 64 lanes, so all scalar logic is wave-uniform: no LDS, no barrier, no
// ds_bpermute in here.  VT = vector storage type, AT = accumulation type of the
// dot products (float/float in production - the reference's vectors and torch.dot are float32 -
// double/double in the known-answer test); line-search scalars are always double (Python floats
// in the reference).  The (s, y) history rows are streamed with a 4-deep register prefetch ring,
// from LDS (single-launch fit) or HBM/L2 (one launch per closure round).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

namespace mvfit {

constexpr int LB_EPL = 2;          // elements per lane
constexpr int LB_D = 96;           // row stride of every optimiser vector (>= D); lanes >= LB_D / LB_EPL idle
constexpr int LB_HIST = 100;
constexpr int LB_PD = 4;           // history prefetch depth (rows in flight)

enum LbPhase : int { PH_STEP_START = 0, PH_LS_FIRST = 1, PH_LS_BRACKET = 2, PH_LS_ZOOM = 3 };

struct LbOpts {
    double lr, tol_grad, tol_change, ftol, gtol;
    int max_iter, max_eval, history, maxiters, num_stages;
    int nseg;            // parameter tensors taking part in the gtol test (compact index ranges)
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
    double br0, br1, bf0, bf1, bgtd0, bgtd1;     // line-search bracket (no dynamic indexing: stays in registers)
};
#define LB_SEL(S, f, i) ((i) ? (S).f##1 : (S).f##0)
#define LB_SET(S, i, t_, f_, g_) do { if (i) { (S).br1 = (t_); (S).bf1 = (f_); (S).bgtd1 = (g_); } \
                                      else { (S).br0 = (t_); (S).bf0 = (f_); (S).bgtd0 = (g_); } } while (0)

template <typename VT>
struct LbVecs {            // lane-distributed working vectors
    VT x[LB_EPL], d[LB_EPL], g[LB_EPL], pg[LB_EPL], gprev[LB_EPL], bg0[LB_EPL], bg1[LB_EPL];
};
constexpr int LB_NVEC = 7;

// History ring: y = dirs, s = stps, row stride LB_D; ro = 1/(y.s).  Generic pointers (LDS or global).
template <typename VT, typename AT>
struct LbHist {
    VT* dirs;      // [LB_HIST][LB_D]
    VT* stps;      // [LB_HIST][LB_D]
    AT* ro;        // [LB_HIST]
};

template <typename VT, typename AT>
__device__ __forceinline__ AT vdot(const VT* a, const VT* b) {
    AT s = (AT)0;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) s = fma((AT)a[e], (AT)b[e], s);
    return wave64_sum(s);
}
template <typename VT>
__device__ __forceinline__ double vmaxabs(const VT* a) {
    VT s = (VT)0;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) s = fmax(s, fabs(a[e]));
    return (double)wave64_max(s);
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

template <typename VT, typename AT>
struct LbRow { VT s[LB_EPL], y[LB_EPL]; AT ro; };

template <typename VT, typename AT>
__device__ __forceinline__ void lb_load_row(LbRow<VT, AT>& R, const LbHist<VT, AT>& Hh, int slot, int lane) {
    if (LB_EPL * lane < LB_D) {
        const VT* ps = Hh.stps + slot * LB_D + LB_EPL * lane;
        const VT* py = Hh.dirs + slot * LB_D + LB_EPL * lane;
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { R.s[e] = ps[e]; R.y[e] = py[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { R.s[e] = (VT)0; R.y[e] = (VT)0; }
    }
    R.ro = Hh.ro[slot];
}

// direction d = -H g by the two-loop recursion (lbfgs_ls.py:336-358), same operation order as the
// reference: al_i = (s_i . q) ro_i ; q -= al_i y_i ; r = q H ; be_i = (y_i . r) ro_i ; r += (al_i - be_i) s_i
template <typename VT, typename AT>
__device__ void lb_two_loop(const LbState& S, const LbVecs<VT>& V, const LbHist<VT, AT>& Hh, VT* dout, int lane) {
    const int n = __builtin_amdgcn_readfirstlane(S.hist_len), head = __builtin_amdgcn_readfirstlane(S.hist_head);
    VT q[LB_EPL];
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) q[e] = -V.g[e];
    AT al0 = (AT)0, al1 = (AT)0;                 // al[i] lives in lane i & 63, register i >> 6
    LbRow<VT, AT> ring[LB_PD];
#pragma unroll
    for (int u = 0; u < LB_PD; ++u) {
        const int i = n - 1 - u;
        if (i >= 0) lb_load_row(ring[u], Hh, (head + i) % LB_HIST, lane);
    }
    for (int base = n - 1; base >= 0; base -= LB_PD) {
#pragma unroll
        for (int u = 0; u < LB_PD; ++u) {
            const int i = base - u;
            if (i >= 0) {
                const AT a = vdot<VT, AT>(ring[u].s, q) * ring[u].ro;
                if (lane == (i & 63)) { if (i < 64) al0 = a; else al1 = a; }
#pragma unroll
                for (int e = 0; e < LB_EPL; ++e) q[e] = (VT)fma(-a, (AT)ring[u].y[e], (AT)q[e]);
                const int nx = i - LB_PD;
                if (nx >= 0) lb_load_row(ring[u], Hh, (head + nx) % LB_HIST, lane);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) q[e] = (VT)((AT)q[e] * (AT)S.H);
#pragma unroll
    for (int u = 0; u < LB_PD; ++u)
        if (u < n) lb_load_row(ring[u], Hh, (head + u) % LB_HIST, lane);
    for (int base = 0; base < n; base += LB_PD) {
#pragma unroll
        for (int u = 0; u < LB_PD; ++u) {
            const int i = base + u;
            if (i < n) {
                const AT be = vdot<VT, AT>(ring[u].y, q) * ring[u].ro;
                const AT a = lane_read((i < 64) ? al0 : al1, i & 63);
                const AT c = a - be;
#pragma unroll
                for (int e = 0; e < LB_EPL; ++e) q[e] = (VT)fma(c, (AT)ring[u].s[e], (AT)q[e]);
                const int nx = i + LB_PD;
                if (nx < n) lb_load_row(ring[u], Hh, (head + nx) % LB_HIST, lane);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) dout[e] = q[e];
}

// Consume (f_new, gnew) of the last closure call, emit the next trial point into xt.
// Returns with S.status == 1 when all stages are finished (xt = final x).
// lane = threadIdx & 63; element e of this lane is flat index LB_EPL * lane + e.
template <typename VT, typename AT>
__device__ void lbfgs_advance(LbState& S, LbVecs<VT>& V, const LbHist<VT, AT>& Hh, const LbOpts& O,
                              double f_new, const VT* gnew, VT* xt, int lane, double* stage_final) {
    const double c1 = 1e-4, c2 = 0.9;
    const int max_ls = 25;
    double gtd_new = 0.0;

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
    for (int e = 0; e < LB_EPL; ++e) V.g[e] = gnew[e];
    S.cur_evals = 1;
    if (vmaxabs(V.g) <= O.tol_grad) goto L_step_return;
    S.n = 0;

L_iter:
    S.n += 1;
    S.n_iter += 1;
    S.n_lbfgs += 1;
    if (S.n_iter == 1) {                                              // :312-317
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) V.d[e] = -V.g[e];
        S.hist_len = 0;
        S.hist_head = 0;
        S.H = 1.0;
    } else {                                                          // :318-358
        VT y[LB_EPL], s[LB_EPL];
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            y[e] = V.g[e] - V.pg[e];
            s[e] = (VT)((AT)V.d[e] * (AT)S.t);
        }
        const AT ys = vdot<VT, AT>(y, s);
        if (ys > (AT)1e-10) {
            if (S.hist_len == O.history) S.hist_head = (S.hist_head + 1) % LB_HIST;
            else S.hist_len += 1;
            const int slot = (S.hist_head + S.hist_len - 1) % LB_HIST;
            if (LB_EPL * lane < LB_D) {
#pragma unroll
                for (int e = 0; e < LB_EPL; ++e) {
                    Hh.dirs[slot * LB_D + LB_EPL * lane + e] = y[e];
                    Hh.stps[slot * LB_D + LB_EPL * lane + e] = s[e];
                }
            }
            if (lane == 0) Hh.ro[slot] = (AT)1 / ys;
            S.H = (double)(ys / vdot<VT, AT>(y, y));
            wave_lds_fence();                         // history is re-read by all lanes of this wave
        }
        lb_two_loop<VT, AT>(S, V, Hh, V.d, lane);
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) V.pg[e] = V.g[e];                // :360-364
    S.prev_loss = S.loss;
    if (S.n_iter == 1) {                                              // :370-373
        AT asum = (AT)0;
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) asum += fabs((AT)V.g[e]);
        asum = wave64_sum(asum);
        S.t = fmin(1.0, 1.0 / (double)asum) * O.lr;
    } else {
        S.t = O.lr;
    }
    S.gtd = (double)vdot<VT, AT>(V.g, V.d);                           // :376
    if (S.gtd > -O.tol_change) goto L_step_return;                    // :379-380
    // ---- _strong_Wolfe entry (:43-53) ----
    S.d_norm = vmaxabs(V.d);
    S.f0 = S.loss;
    S.ls_evals = 0;
    S.phase = PH_LS_FIRST;
    goto L_emit_trial;

L_ls_first:
    S.ls_evals = 1;
    gtd_new = (double)vdot<VT, AT>(gnew, V.d);
    S.t_prev = 0.0; S.f_prev = S.f0; S.gtd_prev = S.gtd;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) V.gprev[e] = V.g[e];
    S.ls_done = 0;
    S.ls_it = 0;
    goto L_bracket_check;

L_ls_bracket:                                                         // :90-93
    S.ls_evals += 1;
    gtd_new = (double)vdot<VT, AT>(gnew, V.d);
    S.ls_it += 1;

L_bracket_check:                                                      // :54-93
    if (S.ls_it < max_ls) {
        bool two_point = false;
        if (f_new > (S.f0 + c1 * S.t * S.gtd) || (S.ls_it > 1 && f_new >= S.f_prev)) {
            two_point = true;                                         // :56-61
        } else if (fabs(gtd_new) <= -c2 * S.gtd) {                    // :63-69
            S.br0 = S.t; S.bf0 = f_new; S.bgtd0 = gtd_new;
#pragma unroll
            for (int e = 0; e < LB_EPL; ++e) V.bg0[e] = gnew[e];
            S.nbr = 1;
            S.ls_done = 1;
            goto L_bracket_end;
        } else if (gtd_new >= 0.0) {                                  // :71-76
            two_point = true;
        }
        if (two_point) {
            S.br0 = S.t_prev; S.br1 = S.t;
            S.bf0 = S.f_prev; S.bf1 = f_new;
            S.bgtd0 = S.gtd_prev; S.bgtd1 = gtd_new;
#pragma unroll
            for (int e = 0; e < LB_EPL; ++e) { V.bg0[e] = V.gprev[e]; V.bg1[e] = gnew[e]; }
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
            for (int e = 0; e < LB_EPL; ++e) V.gprev[e] = gnew[e];
        }
        S.phase = PH_LS_BRACKET;
        goto L_emit_trial;
    }

L_bracket_end:
    if (S.ls_it == max_ls) {                                          // :96-100
        S.br0 = 0.0; S.br1 = S.t;
        S.bf0 = S.f0; S.bf1 = f_new;
        S.bgtd0 = S.gtd; S.bgtd1 = gtd_new;
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { V.bg0[e] = V.g[e]; V.bg1[e] = gnew[e]; }
        S.nbr = 2;
    }
    S.insuf = 0;
    if (S.bf0 <= (S.nbr == 2 ? S.bf1 : S.bf0)) { S.low = 0; S.high = 1; } else { S.low = 1; S.high = 0; }

L_zoom_check:                                                         // :108-130
    if (!S.ls_done && S.ls_it < O.max_iter) {
        double tt = lb_cubic(S.br0, S.bf0, S.bgtd0, S.br1, S.bf1, S.bgtd1, false, 0, 0);
        double bmax = fmax(S.br0, S.br1), bmin = fmin(S.br0, S.br1);
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
    gtd_new = (double)vdot<VT, AT>(gnew, V.d);
    S.ls_it += 1;
    if (f_new > (S.f0 + c1 * S.t * S.gtd) || f_new >= LB_SEL(S, bf, S.low)) {
        int h = S.high;
        LB_SET(S, h, S.t, f_new, gtd_new);
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { if (h == 0) V.bg0[e] = gnew[e]; else V.bg1[e] = gnew[e]; }
        if (S.bf0 <= S.bf1) { S.low = 0; S.high = 1; } else { S.low = 1; S.high = 0; }
    } else {
        if (fabs(gtd_new) <= -c2 * S.gtd) {
            S.ls_done = 1;
        } else if (gtd_new * (LB_SEL(S, br, S.high) - LB_SEL(S, br, S.low)) >= 0.0) {
            int h = S.high, l = S.low;
            { const double t_ = LB_SEL(S, br, l), f_ = LB_SEL(S, bf, l), g_ = LB_SEL(S, bgtd, l); LB_SET(S, h, t_, f_, g_); }
#pragma unroll
            for (int e = 0; e < LB_EPL; ++e) { if (h == 0) V.bg0[e] = V.bg1[e]; else V.bg1[e] = V.bg0[e]; }
        }
        int l = S.low;
        LB_SET(S, l, S.t, f_new, gtd_new);
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { if (l == 0) V.bg0[e] = gnew[e]; else V.bg1[e] = gnew[e]; }
    }
    if (fabs(S.br1 - S.br0) * S.d_norm < O.tol_change) goto L_ls_return;
    goto L_zoom_check;

L_ls_return:                                                          // :163-167, :393-399
    {
        int l = S.low;
        S.loss = LB_SEL(S, bf, l);
        S.t = LB_SEL(S, br, l);
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            V.g[e] = (l == 0) ? V.bg0[e] : V.bg1[e];
            V.x[e] = (VT)fma((AT)S.t, (AT)V.d[e], (AT)V.x[e]);
        }
        S.cur_evals += S.ls_evals;
    }
    if (S.n == O.max_iter) goto L_step_return;                        // :419-434
    if (S.cur_evals >= O.max_eval) goto L_step_return;
    if (vmaxabs(V.g) <= O.tol_grad) goto L_step_return;
    if (S.d_norm * fabs(S.t) <= O.tol_change) goto L_step_return;     // max|d t| == max|d| |t|
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
                    VT m = (VT)-INFINITY;
#pragma unroll
                    for (int e = 0; e < LB_EPL; ++e) {
                        const int ix = LB_EPL * lane + e;
                        if (ix >= O.seg_lo[sgi] && ix < O.seg_hi[sgi]) m = fmax(m, gnew[e]);
                    }
                    m = wave64_max(m);
                    if (!(fabs((double)m) < O.gtol)) all_small = false;
                }
                if (all_small) stop = true;
            }
            if (!stop) { S.outer_prev = loss_out; S.has_outer_prev = 1; }
        }
        S.outer_n += 1;
        if (stop || S.outer_n >= O.maxiters) {
            if (lane == 0) stage_final[S.stage] = S.has_outer_prev ? S.outer_prev : (double)NAN;
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
        for (int e = 0; e < LB_EPL; ++e) xt[e] = V.x[e];
        return;
    }

L_emit_trial:                                                         // _directional_evaluate :249-254
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) xt[e] = (VT)fma((AT)S.t, (AT)V.d[e], (AT)V.x[e]);
    return;
}

}  // namespace mvfit
