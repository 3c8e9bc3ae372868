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
// bit-identical in all 64 lanes, so all scalar logic is wave-uniform.  T = storage and
// accumulation type (float in production - the reference's vectors and torch.dot are float32 -
// double in the known-answer test); line-search scalars are always double (Python floats in the
// reference).
//
// The two-loop recursion (lbfgs_ls.py:336-358) is the one part that is NOT wave-local.  Written
// literally it is 2 m dependent (dot -> axpy) steps and every dot is a ~110-cycle cross-lane
// reduction (measured, tests/microbench/lat.hip).  It is evaluated here in Gram form - same
// products, the sums regrouped - with G[a][b] = s_a . y_b kept per problem:
//     first loop   al_i = ro_i (s_i.q_i),  q_i = -g - sum_{j>i} al_j y_j
//              =>  al_i = ro_i (b_i - sum_{j>i} al_j G[i][j]),  b_i = s_i.(-g)
//     second loop  be_i = ro_i (y_i.r_i),  r_i = H q_0 + sum_{j<i} (al_j - be_j) s_j
//              =>  be_i = ro_i (e_i + sum_{j<i} c_j G[j][i]),  e_i = y_i.(H q_0),  c_j = al_j - be_j
// The m dots b_i / e_i and the two mat-vecs (q_0, d) are independent and run on all waves of the
// workgroup; the two triangular recurrences run column-oriented on wave 0 (lane k owns the
// running value of index k; per step one v_readlane + one FMA, G rows prefetched 8 deep).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

namespace mvfit {

constexpr int LB_EPL = 2;          // elements per lane
constexpr int LB_D = 96;           // row stride of every optimiser vector (>= D); lanes >= LB_D / LB_EPL idle
constexpr int LB_HIST = 100;
constexpr int LB_PD = 8;  // Gram rows in flight in the recurrences
constexpr int LB_GS = 104;         // row stride of the Gram matrices (columns indexed by history slot)
constexpr int LB_GPAD = 2 * LB_PD;                  // pad rows on both sides (prefetch runs up to 2 LB_PD - 1 rows past the end)
constexpr int LB_GROWS = 2 * LB_HIST + 2 * LB_GPAD; // rows: pad | slots 0..99 | slots 0..99 again | pad
constexpr int LB_GSIZE = LB_GROWS * LB_GS;          // elements of one Gram matrix

enum LbPhase : int { PH_STEP_START = 0, PH_LS_FIRST = 1, PH_LS_BRACKET = 2, PH_LS_ZOOM = 3, PH_DIRECTION = 4 };

struct LbOpts {
    double lr, tol_grad, tol_change, ftol, gtol;
    int max_iter, max_eval, history, maxiters, num_stages;
    int nseg;            // parameter tensors taking part in the gtol test (compact index ranges)
    int seg_lo[8], seg_hi[8];
    // Opt-in (MVFIT_F_REUSE_OUTER_VALUE), off in every parity test: LBFGS.step() opens with a closure call
    // (lbfgs_ls.py:279-283) at the point the previous step() of the same stage ended on, whose loss and gradient the
    // optimiser still holds - the accepted line-search point (:393-399).  With the flag the device feeds those back
    // instead of evaluating again: same iterates, same eval accounting, 8-10 % fewer closure evaluations.
    int reuse_outer, pad_;
};

// Per-problem scalar state (lives in global memory between launches).
struct LbState {
    int phase, status, stage, outer_n;
    int n_iter, n, cur_evals;
    int hist_len, hist_head;
    int ls_it, ls_evals, ls_done, low, high, insuf, nbr;
    int has_outer_prev;
    int n_closure, n_lbfgs;
    int ins_slot;        // history slot written by the current iteration (-1: pair rejected)
    int reuse_ok, pad0_, pad1_, pad2_;  // (loss, g) are the closure's values at x: the opening closure call of the next step() may be skipped
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

// History ring: y = dirs, s = stps, row stride LB_D; ro = 1/(y.s).  dirs/stps/ro may be LDS or
// global (generic pointers).  Gram matrices (global), rows and columns indexed by history slot,
// pre-scaled and exactly zero outside their triangle so that a recurrence step is one FMA:
//     gcol[i][k] = ro_k (s_k . y_i)  for k older than i, else 0      (first loop, row = newer pair)
//     grow[i][k] = ro_k (s_i . y_k)  for k newer than i, else 0      (second loop, row = older pair)
// Every row is stored twice (slot and slot + 100) so that "age order" is a linear walk without a
// ring wrap; LB_PD pad rows on both sides keep the prefetch in bounds.
template <typename T>
struct LbHist {
    T* dirs;       // [LB_HIST][ld]
    T* stps;       // [LB_HIST][ld]
    T* ro;         // [LB_HIST]
    T* grow;       // [LB_GROWS][LB_GS]   (two-loop form only)
    T* gcol;       // [LB_GROWS][LB_GS]
    // compact form (lb_direction_compact):
    //   ys[i]   = s_i . y_i by history slot (the diagonal D of the compact representation; ro = 1 / ys)
    //   rinv    = R^-1, R_ij = s_i . y_j for pair i not newer than pair j (upper triangular in age order), stored by SLOT in
    //             packed symmetric form: the entry of the unordered slot pair {r, c} lives at tri(max) + min and holds
    //             (R^-1)_(older, newer).  Exactly one of the two orders is live for any two live pairs, whatever the ring
    //             head: neither accepting a pair (its column overwrites the evicted pair's star) nor evicting one moves data.
    T* ys = nullptr;
    T* rinv = nullptr;
    int ld = LB_D;  // row stride of dirs / stps
};
constexpr int LB_RPACK = (LB_HIST * (LB_HIST + 1) / 2 + 3) / 4 * 4;        // elements of the packed R^-1 (5052)
__device__ __forceinline__ int lb_tri(int r, int c) { const int hi = max(r, c), lo = min(r, c); return ((hi * (hi + 1)) >> 1) + lo; }

// LDS scratch of the block-wide direction computation
template <typename T>
struct LbWork {
    T qv[LB_D];            // -g, then (two-loop form) r_0 = H q_0
    T dv[LB_D];            // direction
    T tv[LB_D];            // compact form: t = Y^T w - q
    T bvec[128];           // two-loop: b_i, then e_i by age ; compact: p = S q by slot
    T alpha[128];          // two-loop: al_i by age          ; compact: w = R^-1 p by slot
    T cvec[128];           // two-loop: c_i by age           ; compact: a = R^-T z by slot
    T gnew[128];           // s_a . y_new by slot (insert pass; compact: u)
    T zv[128];             // compact: z = D w + gamma Y t by slot
    union {
        T part[8][LB_D];       // per-wave partial sums of the history mat-vecs
        T part2[2][4][128];    // compact: partial sums of the two triangular products, [w | u][column group][slot]
    };
    int n, head, ins_slot, need_dir;
    T Hdiag;
};

template <typename T>
__device__ __forceinline__ T vdot(const T* a, const T* b) {
    T s = (T)0;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) s = fma(a[e], b[e], s);
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

// dots of history rows against LDS vectors: 16 lanes per row (6 elements each), NT/16 rows per pass.
//   which = 0: b_a = s_a . qv  and, if a pair was inserted, W.gnew[slot_a] = s_a . y_new
//   which = 1: e_a = y_a . qv
template <typename T, int NT>
__device__ void lb_row_dots(const LbHist<T>& Hh, LbWork<T>& W, int which, int tid) {
    const int n = W.n, head = W.head, ins = W.ins_slot;
    const int l16 = tid & 15;
    const bool two = which == 0 && ins >= 0;
    T qv[6], yn[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) { qv[e] = W.qv[6 * l16 + e]; yn[e] = (T)0; }
    if (two) {
#pragma unroll
        for (int e = 0; e < 6; ++e) yn[e] = Hh.dirs[ins * Hh.ld + 6 * l16 + e];
    }
    for (int a = tid >> 4; a < n; a += NT / 16) {
        int slot = head + a;
        slot = slot >= LB_HIST ? slot - LB_HIST : slot;
        const T* row = (which == 0 ? Hh.stps : Hh.dirs) + slot * Hh.ld + 6 * l16;
        T d0 = (T)0, d1 = (T)0;
#pragma unroll
        for (int e = 0; e < 6; ++e) { const T v = row[e]; d0 = fma(v, qv[e], d0); d1 = fma(v, yn[e], d1); }
        d0 = row16_sum(d0);
        if (two) d1 = row16_sum(d1);
        if (l16 == 0) { W.bvec[a] = d0; if (two) W.gnew[slot] = d1; }
    }
}

// Window of Gram rows staged in LDS by the step kernel (direct global -> LDS loads, issued long before the recurrence
// runs): physical rows [row0, row0 + LB_GW_ROWS) of ONE matrix; a row of the matrix is its own cache line and cold after
// a launch boundary, which costs the recurrence ~100 cycles per step from global and ~55 from LDS
// (tests/microbench/recur.hip).
constexpr int LB_GW_ROWS = LB_HIST + 3 + 4 * LB_PD;
constexpr int LB_GW_BYTES = ((LB_GW_ROWS * LB_GS * 4 + 1023) / 1024) * 1024;
struct LbGramLds {
    float* buf;          // LDS, LB_GW_BYTES; null: recurrences read the matrices from global memory
    int row0;            // physical row of buf row 0
    int nrows;           // rows staged
};

// all threads: request the window of matrix M that starts at physical row row0 (16-byte words, 1 KiB per wave load)
template <int NT>
__device__ __forceinline__ void lb_gram_dma(const float* M, int row0, float* lds, int nrows, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const float4* src = reinterpret_cast<const float4*>(M + (size_t)row0 * LB_GS);
    const int last = (LB_GROWS - row0) * (LB_GS / 4) - 1;          // stay inside the matrix
    const int n4 = nrows * (LB_GS / 4);
    for (int c = wave; c * 64 < n4; c += NT / 64) {
        const int i = min(c * 64 + lane, last);
        __builtin_amdgcn_global_load_lds(src + i, reinterpret_cast<float4*>(lds) + c * 64, 16, 0, 0);
    }
}

// Gram maintenance after the row dots of an inserted pair t: rows t and column t of both matrices
// (see LbHist).  Thread a < 100 owns history slot a.
template <typename T, int NT>
__device__ void lb_gram_insert(const LbHist<T>& Hh, LbWork<T>& W, int tid, const LbGramLds& GL) {
    const int n = W.n, head = W.head, t = W.ins_slot;
    // the staged window of gcol (first recurrence) gets the same row / column of the new pair
    auto lds_put = [&](int r, int col, T val) {          // r: row index relative to the first non-pad row
        const int lr = LB_GPAD + r - GL.row0;
        if (sizeof(T) == 4 && GL.buf && lr >= 0 && lr < LB_GW_ROWS) reinterpret_cast<T*>(GL.buf)[lr * LB_GS + col] = val;
    };
    for (int a = tid; a < LB_HIST; a += NT) {
        int age = a - head;
        age = age < 0 ? age + LB_HIST : age;                     // age of slot a (valid if < n)
        const bool older = age < n && a != t;                    // live and older than the new pair
        const T g = older ? W.gnew[a] : (T)0;                    // s_a . y_t
        const T c = older ? Hh.ro[a] * g : (T)0;                 // gcol[t][a] = ro_a (s_a . y_t)
        const T r = older ? Hh.ro[t] * g : (T)0;                 // grow[a][t] = ro_t (s_a . y_t)
        T* gc = Hh.gcol + LB_GPAD * LB_GS;
        T* gr = Hh.grow + LB_GPAD * LB_GS;
        gc[t * LB_GS + a] = c;              gc[(t + LB_HIST) * LB_GS + a] = c;
        lds_put(t, a, c); lds_put(t + LB_HIST, a, c); lds_put(a, t, (T)0); lds_put(a + LB_HIST, t, (T)0);
        gr[t * LB_GS + a] = (T)0;           gr[(t + LB_HIST) * LB_GS + a] = (T)0;
        gc[a * LB_GS + t] = (T)0;           gc[(a + LB_HIST) * LB_GS + t] = (T)0;
        gr[a * LB_GS + t] = r;              gr[(a + LB_HIST) * LB_GS + t] = r;
    }
}

// out[e] = (in[e] + sign * sum_j coef[j] * rows_j[e]) * scale, j split over the waves
template <typename T, int NT>
__device__ void lb_matvec(const T* rows, const T* coef, T sign, T scale, const T* in, T* outvec, LbWork<T>& W, int tid) {
    constexpr int NW = NT / 64;
    const int n = W.n, head = W.head;
    const int wave = tid >> 6, lane = tid & 63;
    if (LB_EPL * lane < LB_D) {
        T a0 = (T)0, a1 = (T)0;
        for (int j = wave; j < n; j += NW) {
            int slot = head + j;
            slot = slot >= LB_HIST ? slot - LB_HIST : slot;
            const T c = coef[j];
            a0 = fma(c, rows[slot * LB_D + 2 * lane], a0);
            a1 = fma(c, rows[slot * LB_D + 2 * lane + 1], a1);
        }
        W.part[wave][2 * lane] = a0;
        W.part[wave][2 * lane + 1] = a1;
    }
    __syncthreads();
    for (int e = tid; e < LB_D; e += NT) {
        T s = (T)0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += W.part[w][e];
        outvec[e] = (in[e] + sign * s) * scale;
    }
    __syncthreads();
}

// One triangular recurrence on wave 0: x_k -= x_i * M[i][k] for i in age order (DESC: newest ->
// oldest over gcol; else oldest -> newest over grow).  Lane k owns ages k and k + 64; when the
// loop is done lane k holds the solution component of age k (al_k, resp. c_k = al_k - be_k).
// Per step: one prefetched row element per owned age, one v_readlane, one FMA.
template <typename T, bool DESC, bool TWO, bool LDSM = false>
__device__ __forceinline__ void lb_recur_loop(const T* Mg, int n, int head, int lane, T& x0, T& x1) {
    typedef typename std::conditional<LDSM, __attribute__((address_space(3))) const T*, const T*>::type MP;
    MP M = (MP)Mg;
    int s0 = head + lane, s1 = head + lane + 64;
    s0 = s0 >= LB_HIST ? s0 - LB_HIST : s0;
    s1 = s1 >= LB_HIST ? s1 - LB_HIST : s1;
    s1 = s1 >= LB_HIST ? s1 - LB_HIST : s1;
    // lanes that own no live age read column LB_HIST (never written, zero in every row): their x stays 0
    if (lane >= n) s0 = LB_HIST;
    if (lane + 64 >= n) s1 = LB_HIST;
    // row of age i is physical row LB_GPAD + head + i (doubled layout: no wrap for i < 100)
    MP p0 = M + (LB_GPAD + head + (DESC ? n - 1 : 0)) * LB_GS + s0;
    MP p1 = M + (LB_GPAD + head + (DESC ? n - 1 : 0)) * LB_GS + s1;
    constexpr int RS = DESC ? -LB_GS : LB_GS;
    T g0[LB_PD], g1[LB_PD];
    // Drain this wave's earlier global stores (Gram insertion) first: with stores still counted in vmcnt the compiler
    // cannot tell which of the row loads below have returned and waits for ALL of them at the top of every group of
    // LB_PD steps - one full memory latency per group (~110 cycles per step instead of ~25, ISA-verified).
    if (!LDSM) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
    for (int u = 0; u < LB_PD; ++u) { g0[u] = p0[u * RS]; g1[u] = TWO ? p1[u * RS] : (T)0; }
    // Steps st >= n of the last group run unguarded: the broadcast then reads a lane that owns no live
    // age (index >= n, or negative & 127), whose x is exactly 0, so x is left unchanged.
    for (int base = 0; base < n; base += LB_PD) {
        p0 += LB_PD * RS;
        p1 += LB_PD * RS;
#pragma unroll
        for (int u = 0; u < LB_PD; ++u) {
            const int st = base + u;
            const int i = (DESC ? n - 1 - st : st) & 127;
            const T v = TWO ? lane_read(i < 64 ? x0 : x1, i & 63) : lane_read(x0, i & 63);
            x0 = fma(-v, g0[u], x0);
            if (TWO) x1 = fma(-v, g1[u], x1);
            g0[u] = p0[u * RS];                 // row st + LB_PD (pad rows keep this in bounds)
            if (TWO) g1[u] = p1[u * RS];
        }
    }
}

template <typename T, bool FIRST>
__device__ void lb_recurrence(const LbHist<T>& Hh, LbWork<T>& W, int lane, const LbGramLds& GL) {
    const int n = __builtin_amdgcn_readfirstlane(W.n), head = __builtin_amdgcn_readfirstlane(W.head);
    const int k0 = lane, k1 = lane + 64;
    int s0 = head + k0, s1 = head + k1;
    s0 = s0 >= LB_HIST ? s0 - LB_HIST : s0;
    s1 = s1 >= LB_HIST ? s1 - LB_HIST : s1;
    s1 = s1 >= LB_HIST ? s1 - LB_HIST : s1;
    T x0 = (T)0, x1 = (T)0;
    if (FIRST) {       // r'_k = ro_k b_k
        if (k0 < n) x0 = Hh.ro[s0] * W.bvec[k0];
        if (k1 < n) x1 = Hh.ro[s1] * W.bvec[k1];
    } else {           // w_k = al_k - ro_k e_k
        if (k0 < n) x0 = W.alpha[k0] - Hh.ro[s0] * W.bvec[k0];
        if (k1 < n) x1 = W.alpha[k1] - Hh.ro[s1] * W.bvec[k1];
    }
    if (sizeof(T) == 4 && GL.buf) {
        // virtual base so that "physical row r" of the loop lands on window row r - row0
        const T* Mv = reinterpret_cast<const T*>(GL.buf) - (ptrdiff_t)GL.row0 * LB_GS;
        if (n > 64) lb_recur_loop<T, FIRST, true, true>(Mv, n, head, lane, x0, x1);
        else lb_recur_loop<T, FIRST, false, true>(Mv, n, head, lane, x0, x1);
    } else {
        if (n > 64) lb_recur_loop<T, FIRST, true>(FIRST ? Hh.gcol : Hh.grow, n, head, lane, x0, x1);
        else lb_recur_loop<T, FIRST, false>(FIRST ? Hh.gcol : Hh.grow, n, head, lane, x0, x1);
    }
    T* out = FIRST ? W.alpha : W.cvec;
    if (k0 < n) out[k0] = x0;
    if (k1 < n) out[k1] = x1;
}

// ---------------------------------------------------------------------------------------------------------
// d = H (-g) in the compact (Byrd-Nocedal-Schnabel 1994, eq. 3.1) form of the same L-BFGS matrix the two-loop recursion
// applies (lbfgs_ls.py:336-358), H_0 = gamma I, q = -g:
//     H q = gamma q + S a - gamma Y w,   w = R^-1 S^T q,   a = R^-T ((D + gamma Y^T Y) w - gamma Y^T q)
// with R_ij = s_i . y_j (i not newer than j), D = diag(s_i . y_i).  With t = Y w - q this is
//     p = S^T q ;  w = R^-1 p ;  t = Y w - q ;  z = D w + gamma Y^T t ;  a = R^-T z ;  d = S a - gamma t
// - four passes over the history and two triangular mat-vecs, every one of them parallel over the whole workgroup; the
// two-loop form's 2 m dependent steps on one wave are gone.  R^-1 is maintained explicitly (LbHist::rinv): accepting a pair
// borders it with one column, [R u; 0 rho]^-1 = [R^-1, -R^-1 u / rho; 0, 1 / rho], which is the same triangular product
// with u = S^T y_new as right-hand side; evicting the oldest pair drops its row and column and leaves the rest untouched.
// Same direction as the two-loop recursion up to rounding: the float64 instantiation follows the reference optimiser's
// trajectories to 1e-7 (tests/test_gpu_lbfgs.py, form 'compact'); in float32 on the real objective the two forms are
// equally far from each other as from their own float64 runs (tools/lbfgs_direction_study.py).
// All sums have a fixed association (independent of timing; of the launch geometry only through NT).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lb_age(int slot, int head) { const int a = slot - head; return a < 0 ? a + LB_HIST : a; }

// (All four primitives are written branch-free with clamped addresses and zeroed coefficients: every load of a thread is
// issued before the first use - a phase costs one LDS round trip plus its instruction count, ~8 cycles per instruction and
// thread with the 8 waves of a workgroup on 4 SIMDs - instead of one round trip per row / column.)

// history rows . vector: 4 lanes per row (interleaved 4-element chunks), all LB_HIST slots in NT / 4 rows per pass.
//   FIRST: p[slot] = s_slot . q  and, with a freshly inserted pair t, u[slot] = s_slot . y_t (0 for slot t)
//   else : z[slot] = ys[slot] w[slot] + gamma (y_slot . t)
// four consecutive elements of a 16-byte aligned row position as ONE LDS read (rows, work vectors and the history stride are
// 16-byte multiples by construction; the compiler cannot see it through the run-time stride and would emit four reads)
template <typename T>
__device__ __forceinline__ void lb_load4(const T* p, T (&o)[4]) {
    if constexpr (sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = p[j];
    }
}

template <typename T, int NT, bool FIRST>
__device__ __forceinline__ void lb_cmp_rowdots(const LbHist<T>& Hh, LbWork<T>& W, int tid) {
    const int n = W.n, head = W.head, t = W.ins_slot, ld = Hh.ld;
    const bool two = FIRST && t >= 0;
    const int q4 = tid & 3, nch = ld >> 2;
    constexpr int NCH = (LB_D / 4 + 3) / 4;               // 4-element chunks per lane (ld <= LB_D)
    constexpr int HB = NCH / 2;                            // two batches of chunks: half the registers of one batch of six
    static_assert(NCH % 2 == 0, "two batches");
    const T* vec = FIRST ? W.qv : W.tv;
    const T* ynew = Hh.dirs + max(t, 0) * ld;
    const T* rows = FIRST ? Hh.stps : Hh.dirs;
    for (int r0 = tid >> 2; r0 < 128; r0 += NT / 4) {
        const int r = min(r0, LB_HIST - 1);
        const bool live = r0 < LB_HIST && lb_age(r, head) < n;
        const T* row = rows + r * ld;
        T a0 = (T)0, a1 = (T)0;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            T vv[HB][4], yn[HB][4], rv[HB][4];
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int ch = q4 + 4 * (hb * HB + k);
                const bool ok = ch < nch;
                const int cc = 4 * min(ch, nch - 1);           // clamped chunk: the surplus lanes re-read the last chunk with zero coefficients
                T v4[4], y4[4] = {(T)0, (T)0, (T)0, (T)0};
                lb_load4(vec + cc, v4);
                if (FIRST) lb_load4(ynew + cc, y4);
                lb_load4(row + cc, rv[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vv[k][j] = ok ? v4[j] : (T)0;
                    yn[k][j] = (ok && two) ? y4[j] : (T)0;
                }
            }
#pragma unroll
            for (int k = 0; k < HB; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) { a0 = fma(rv[k][j], vv[k][j], a0); if (FIRST) a1 = fma(rv[k][j], yn[k][j], a1); }
        }
        a0 += dpp_mov<DPP_XOR1>(a0); a0 += dpp_mov<DPP_XOR2>(a0);
        if (FIRST) { a1 += dpp_mov<DPP_XOR1>(a1); a1 += dpp_mov<DPP_XOR2>(a1); }
        const T ysw = FIRST ? (T)0 : Hh.ys[r] * W.alpha[r];
        if (q4 == 0 && r0 < LB_HIST) {
            if (FIRST) { W.bvec[r] = live ? a0 : (T)0; W.gnew[r] = (live && two && r != t) ? a1 : (T)0; }
            else W.zv[r] = live ? fma(W.Hdiag, a0, ysw) : (T)0;
        }
    }
}

// one triangular product over the packed R^-1, in AGE space (row age a = lane, column age b uniform per wave), partial sums
// per column group g (columns b = g, g + 4, ...: every group gets a quarter of the LIVE columns, whatever the history length):
//   FWD : part2[0][g][slot(a)] = sum_{b in group g, b >= a, b < nE} rinv{a,b} p_b           (w = R^-1 p; nE excludes a pair that is
//         part2[1][g][slot(a)] = the same sum with u_b                                        being inserted: its column is -R^-1 u / rho)
//   else: part2[0][g][slot(a)] = sum_{b in group g, b <= a} rinv{a,b} z_b                    (a = R^-T z)
// Per-lane tables (slot of age, its triangle offset, the vector element), two ages per lane; the column's entries of the
// tables arrive by v_readlane, its use mask (lanes a <= b, resp. a >= b) as a scalar bit field.
// Triangular products with R^-1 (packed by slot) in ONE phase, in age space: item = (row age a, quarter q); the four lanes of a
// row take its live columns with age b = q (mod 4), ascending - so the work is what the live window holds: a row of the
// typical 40-60 pair history is ~12 columns per lane -, and the row total is ((p0 + p1) + p2) + p3 of the four lanes' sums,
// taken with quad broadcasts: no partial sums in LDS, no combine phase, no second barrier.  (The grouping by b mod 4 and the
// order of the four partial sums are those of the round's first version of this product, which ran the four groups on four
// waves and met in LDS: the same bits.)  The lane with q = 0 finishes the row:
//   FWD  w = sum_{a <= b < nE} Rinv(a, b) p_b and, when a pair has just been accepted (slot t, age n - 1), the new column of
//        R^-1: Rinv(a, t) = -ro_t sum_b Rinv(a, b) u_b (u = S y_new), w += Rinv(a, t) p_t; row t itself is (ro_t p_t, ro_t).
//        Nobody reads column t or row t in this phase (its age is outside nE), and the packed entry {slot(a), t} is the one
//        the evicted pair's star occupied: written here, read from the next phase on.  Ages n .. 99 are the dead slots: 0.
//   BWD  c = sum_{b <= a} Rinv(b, a) z_b.
// Loads are issued four columns at a time before their first use.
template <typename T, int NT, bool FWD>
__device__ __forceinline__ void lb_cmp_tri(const LbHist<T>& Hh, LbWork<T>& W, int tid) {
    const int n = W.n, head = W.head;
    const int t = FWD ? W.ins_slot : -1;
    const int nE = (FWD && t >= 0) ? n - 1 : n;            // live columns of this product (the inserted pair is the newest)
    const T* xv = FWD ? W.bvec : W.zv;
    for (int r = LB_HIST + tid; r < 128; r += NT) (FWD ? W.alpha : W.cvec)[r] = (T)0;      // padding slots
    for (int item = tid; item < 4 * 128; item += NT) {
        const int a = item >> 2, q = item & 3;
        const int a0 = __builtin_amdgcn_readfirstlane(a);                   // the wave's first row (NT is a multiple of 64: 16 rows per wave)
        if (a0 >= LB_HIST) break;                                           // uniform: nothing but padding from here on
        const bool live = a < n;
        int sa = head + min(a, LB_HIST - 1);
        sa = sa >= LB_HIST ? sa - LB_HIST : sa;
        T s0 = (T)0, s1 = (T)0;
        if (a0 < n) {                                                       // uniform
            // first column of this lane: the smallest b >= blo with b = q (mod 4); the wave's rows a0 .. a0 + 15 start at
            // different columns in the forward product, so the uniform trip count is taken from the wave's first row
            const int blo = FWD ? a : 0, bend = FWD ? nE : a + 1;
            const int wlo = FWD ? (a0 & ~3) : 0, whi = FWD ? nE : min(a0 + 16, n);      // column range of the whole wave (multiple of 4 at the low end)
            const int trips = (whi - wlo + 3) >> 2;
            for (int k0 = 0; k0 < trips; k0 += 4) {                         // uniform trip count
                T m[4], x[4], u[4];
                bool use[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int bq = wlo + q + 4 * (k0 + j);
                    use[j] = live && bq >= blo && bq < bend;
                    int sb = head + min(max(bq, 0), LB_HIST - 1);
                    sb = sb >= LB_HIST ? sb - LB_HIST : sb;
                    const int hi = max(sa, sb), lo = min(sa, sb);
                    m[j] = Hh.rinv[((hi * (hi + 1)) >> 1) + lo];
                    x[j] = xv[sb];
                    if (FWD) u[j] = W.gnew[sb];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const T mk = use[j] ? m[j] : (T)0;
                    s0 = fma(mk, use[j] ? x[j] : (T)0, s0);
                    if (FWD) s1 = fma(mk, use[j] ? u[j] : (T)0, s1);
                }
            }
            s0 = ((dpp_mov<0x00>(s0) + dpp_mov<0x55>(s0)) + dpp_mov<0xAA>(s0)) + dpp_mov<0xFF>(s0);
            if (FWD) s1 = ((dpp_mov<0x00>(s1) + dpp_mov<0x55>(s1)) + dpp_mov<0xAA>(s1)) + dpp_mov<0xFF>(s1);
        }
        if (q == 0 && a < LB_HIST) {
            if (FWD) {
                T wr = (T)0;
                if (live) {
                    if (sa == t) {
                        wr = Hh.ro[t] * W.bvec[t];
                        Hh.rinv[lb_tri(t, t)] = Hh.ro[t];
                    } else {
                        wr = s0;
                        if (t >= 0) {
                            const T cr = -s1 * Hh.ro[t];
                            Hh.rinv[lb_tri(sa, t)] = cr;
                            wr = fma(cr, W.bvec[t], wr);
                        }
                    }
                }
                W.alpha[sa] = wr;
            } else {
                W.cvec[sa] = live ? s0 : (T)0;
            }
        }
    }
}

// sum_j coef[j] rows_j (a D-vector): 32 lanes per row (4-element chunks), NT / 32 rows per pass, halves of a wave
// combined by v_permlane32_swap, per-wave partials in W.part (the caller sums them after a barrier)
template <typename T, int NT>
__device__ __forceinline__ void lb_cmp_matvec(const T* rows, int ld, const T* coef, LbWork<T>& W, int tid) {
    const int n = W.n, head = W.head;
    const int cl = tid & 31, c = 4 * min(cl, (ld >> 2) - 1);
    const bool cok = 4 * cl < ld;
    constexpr int NWV = (NT / 64) < 8 ? (NT / 64) : 8;     // waves that write W.part: the rows no wave writes are zeroed (the caller adds 8)
    if (NWV < 8) { for (int i = tid; i < (8 - NWV) * LB_D; i += NT) (&W.part[NWV][0])[i] = (T)0; }
    constexpr int RP = NT / 32, NIT = (LB_HIST + RP - 1) / RP;
    T acc[4] = {(T)0, (T)0, (T)0, (T)0};
    for (int i0 = 0; i0 < NIT; i0 += 8) {
        T cf[8], rv[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = (tid >> 5) + (i0 + i) * RP, r = min(r0, LB_HIST - 1);
            const bool ok = (i0 + i) < NIT && r0 < LB_HIST && lb_age(r, head) < n && cok;
            const T cv = coef[r];
            cf[i] = ok ? cv : (T)0;
            lb_load4(rows + r * ld + c, rv[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fma(cf[i], rv[i][j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { T lo, hi; swap_pair<true>(acc[j], lo, hi); acc[j] = lo + hi; }
    if ((tid & 32) == 0 && 4 * cl < LB_D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) W.part[(tid >> 6) & 7][4 * cl + j] = acc[j];
    }
}

// In: W.qv = -g, W.n / head / ins_slot / Hdiag, history rows, ys, ro incl. the freshly inserted pair.  Out: W.dv.
template <typename T, int NT>
__device__ __forceinline__ void lb_direction_compact(const LbHist<T>& Hh, LbWork<T>& W, int tid) {
    const int t = W.ins_slot, head = W.head, n = W.n;
    // ---- p = S^T q (+ u = S^T y_new) ----
    lb_cmp_rowdots<T, NT, true>(Hh, W, tid);
    __syncthreads();
    PH_T(16);
    // ---- w = R^-1 p ; new column of R^-1 ----
    lb_cmp_tri<T, NT, true>(Hh, W, tid);
    __syncthreads();
    PH_T(17);
    // ---- t = Y w - q ----
    lb_cmp_matvec<T, NT>(Hh.dirs, Hh.ld, W.alpha, W, tid);
    __syncthreads();
    for (int e = tid; e < LB_D; e += NT) {
        T sacc = W.part[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) sacc += W.part[k][e];
        W.tv[e] = e < Hh.ld ? sacc - W.qv[e] : (T)0;
    }
    __syncthreads();
    PH_T(18);
    // ---- z = D w + gamma Y^T t ----
    lb_cmp_rowdots<T, NT, false>(Hh, W, tid);
    __syncthreads();
    PH_T(19);
    // ---- a = R^-T z ----
    lb_cmp_tri<T, NT, false>(Hh, W, tid);
    __syncthreads();
    PH_T(20);
    // ---- d = S a - gamma t ----
    lb_cmp_matvec<T, NT>(Hh.stps, Hh.ld, W.cvec, W, tid);
    __syncthreads();
    for (int e = tid; e < LB_D; e += NT) {
        T sacc = W.part[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) sacc += W.part[k][e];
        W.dv[e] = e < Hh.ld ? fma(-W.Hdiag, W.tv[e], sacc) : (T)0;
    }
    __syncthreads();
    PH_T(21);
}

// d = -H g (lbfgs_ls.py:336-358) by the whole workgroup (NT threads, wave 0 = the optimiser wave).
// In: W.qv = -g, W.n/head/ins_slot/Hdiag, history rows incl. the freshly inserted pair.  Out: W.dv.
// (two-loop form: the history rows have stride LB_D here)
template <typename T, int NT>
__device__ __forceinline__ void lb_direction_block(const LbHist<T>& Hh, LbWork<T>& W, int tid, LbGramLds GL = LbGramLds{nullptr, 0, 0}) {
    // the window was requested before the optimiser advanced: it is usable if the ring head is where it was (or one
    // further: the oldest pair was evicted) and the rows the walks touch are inside it; a stage change resets the
    // ring and a head wrap moves the walk far away - then the recurrences read global memory as before
    const bool staged = GL.buf != nullptr;
    if (staged) {
        const int hd = W.head - GL.row0;
        if (hd < 0 || hd > 1 || hd + W.n + 2 + 4 * LB_PD > GL.nrows) GL.buf = nullptr;          // block-uniform
    }
    lb_row_dots<T, NT>(Hh, W, 0, tid);
    if (staged) __builtin_amdgcn_s_waitcnt(0);            // this wave's share of the gcol window has landed
    __syncthreads();
    if (W.ins_slot >= 0) { lb_gram_insert<T, NT>(Hh, W, tid, GL); __syncthreads(); }
    PH_T(16);
    if (tid < 64) lb_recurrence<T, true>(Hh, W, tid, GL);
    __syncthreads();
    PH_T(17);
    if (sizeof(T) == 4 && GL.buf) {
        // the window now takes grow (second recurrence): its rows around the current head, requested here and
        // consumed after the mat-vec and the second row dots
        GL.row0 = W.head;
        GL.nrows = min(W.n, LB_HIST) + 2 + 4 * LB_PD;
        lb_gram_dma<NT>(reinterpret_cast<const float*>(Hh.grow), GL.row0, GL.buf, GL.nrows, tid);
    }
    // r_0 = H (-g - sum_j al_j y_j)
    lb_matvec<T, NT>(Hh.dirs, W.alpha, (T)-1, W.Hdiag, W.qv, W.qv, W, tid);
    PH_T(18);
    lb_row_dots<T, NT>(Hh, W, 1, tid);
    if (GL.buf) __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    PH_T(19);
    if (tid < 64) lb_recurrence<T, false>(Hh, W, tid, GL);
    __syncthreads();
    PH_T(20);
    // d = r_0 + sum_j c_j s_j
    lb_matvec<T, NT>(Hh.stps, W.cvec, (T)1, (T)1, W.qv, W.dv, W, tid);
    PH_T(21);
}

// Consume (f_new, gnew) of the last closure call, emit the next trial point into xt.
// Returns 0 when a trial point was emitted (S.status == 1: all stages finished, xt = final x), or
// 1 when a new search direction is needed: the caller then runs lb_direction_block() with the whole
// workgroup and calls again (S.phase == PH_DIRECTION; f_new / gnew are ignored on that call).
// lane = threadIdx & 63; element e of this lane is flat index LB_EPL * lane + e.
template <typename T>
__device__ __forceinline__ int lbfgs_advance(LbState& S, LbVecs<T>& V, const LbHist<T>& Hh, LbWork<T>& W, const LbOpts& O,
                             double f_new, const T* gnew, T* xt, int lane, double* stage_final, bool virtual_call = false) {
    const double c1 = 1e-4, c2 = 0.9;
    const int max_ls = 25;
    double gtd_new = 0.0;

    if (S.phase == PH_DIRECTION) goto L_have_direction;
    if (!virtual_call) S.n_closure += 1;
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
    PH_T(51);
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
        T y[LB_EPL], s[LB_EPL];
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            y[e] = V.g[e] - V.pg[e];
            s[e] = V.d[e] * (T)S.t;
        }
        const T ys = vdot<T>(y, s);
        S.ins_slot = -1;
        if (ys > (T)1e-10) {
            if (S.hist_len == O.history) S.hist_head = (S.hist_head + 1) % LB_HIST;
            else S.hist_len += 1;
            const int slot = (S.hist_head + S.hist_len - 1) % LB_HIST;
            if (LB_EPL * lane < Hh.ld) {                      // (elements past D are zero; ld is even)
#pragma unroll
                for (int e = 0; e < LB_EPL; ++e) {
                    Hh.dirs[slot * Hh.ld + LB_EPL * lane + e] = y[e];
                    Hh.stps[slot * Hh.ld + LB_EPL * lane + e] = s[e];
                }
            }
            if (lane == 0) { Hh.ro[slot] = (T)1 / ys; if (Hh.ys) Hh.ys[slot] = ys; }
            S.H = (double)(ys / vdot<T>(y, y));
            S.ins_slot = slot;
        }
        if (S.hist_len > 0) {
            // hand the two-loop recursion to the workgroup: W.qv = -g
            if (LB_EPL * lane < LB_D) {
#pragma unroll
                for (int e = 0; e < LB_EPL; ++e) W.qv[LB_EPL * lane + e] = -V.g[e];
            }
            if (lane == 0) { W.n = S.hist_len; W.head = S.hist_head; W.ins_slot = S.ins_slot; W.Hdiag = (T)S.H; }
            S.phase = PH_DIRECTION;
            PH_T(52);
            return 1;
        }
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) V.d[e] = (T)S.H * (-V.g[e]);     // empty history: d = -g H
    }
    goto L_after_direction;
L_have_direction:
    PH_T(53);
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) V.d[e] = (LB_EPL * lane < LB_D) ? W.dv[LB_EPL * lane + e] : (T)0;
L_after_direction:
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) V.pg[e] = V.g[e];                // :360-364
    S.prev_loss = S.loss;
    if (S.n_iter == 1) {                                              // :370-373
        T asum = (T)0;
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) asum += fabs(V.g[e]);
        asum = wave64_sum(asum);
        S.t = fmin(1.0, 1.0 / (double)asum) * O.lr;
    } else {
        S.t = O.lr;
    }
    S.gtd = (double)vdot<T>(V.g, V.d);                                // :376
    if (S.gtd > -O.tol_change) goto L_step_return;                    // :379-380
    // ---- _strong_Wolfe entry (:43-53) ----
    S.d_norm = vmaxabs(V.d);
    S.f0 = S.loss;
    S.ls_evals = 0;
    S.phase = PH_LS_FIRST;
    goto L_emit_trial;

L_ls_first:
    PH_T(48);
    S.ls_evals = 1;
    gtd_new = (double)vdot<T>(gnew, V.d);
    S.t_prev = 0.0; S.f_prev = S.f0; S.gtd_prev = S.gtd;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) V.gprev[e] = V.g[e];
    S.ls_done = 0;
    S.ls_it = 0;
    goto L_bracket_check;

L_ls_bracket:                                                         // :90-93
    S.ls_evals += 1;
    gtd_new = (double)vdot<T>(gnew, V.d);
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
    gtd_new = (double)vdot<T>(gnew, V.d);
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
    PH_T(50);
    {
        int l = S.low;
        S.loss = LB_SEL(S, bf, l);
        S.t = LB_SEL(S, br, l);
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            V.g[e] = (l == 0) ? V.bg0[e] : V.bg1[e];
            V.x[e] = fma((T)S.t, V.d[e], V.x[e]);
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
                    T m = (T)-INFINITY;
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
        S.reuse_ok = O.reuse_outer && !(isnan(loss_out) || isinf(loss_out)) ? 1 : 0;
        if (stop || S.outer_n >= O.maxiters) {
            S.reuse_ok = 0;               // the next stage has other weights: its first closure call is a real one
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
        return 0;
    }

L_emit_trial:                                                         // _directional_evaluate :249-254
    PH_T(54);
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) xt[e] = fma((T)S.t, V.d[e], V.x[e]);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The optimiser state lives in MEMORY between closure rounds (LDS in the fit kernels): LbState + the seven working
// vectors, LANE-MAJOR - lane l's two elements of x, d, g, pg, gprev, bg0, bg1 are one LbVecs<T> (14 words) - so that the
// state machine can work on the memory image IN PLACE: lbfgs_advance(S, V, ...) takes references, and with references
// into LDS it holds nothing but temporaries in registers (as register copies the 60 + 14 words of state, carried
// through its control-flow graph, cost more spills than the rest of the kernel together).
// ---------------------------------------------------------------------------------------------------------
constexpr int LB_LANES = 64;

// The two transitions that make up three quarters of all rounds, as straight-line code on the state in memory - the same
// arithmetic, in the same order, as the walk of lbfgs_advance through the same labels (the general machine carries its 60
// words of state and 14 vector registers through a control-flow graph of ~2000 instructions; a round through it costs
// ~9 k cycles on its one wave).  Nothing is modified unless the transition applies; -1 = not applicable, the caller runs
// lbfgs_advance on the untouched state.  A -DMVFIT_LB_CHECK build runs both and compares every word (lbfgs_round).
//
// (1) the first trial point of a line search satisfies the strong Wolfe conditions and no stopping test fires: accept
//     (lbfgs_ls.py:54-69, 163-167, 393-434), open the next iteration, accept the curvature pair (:318-335) and hand the
//     two-loop recursion to the workgroup.  Returns 1 (direction needed) like lbfgs_advance.
template <typename T>
__device__ __forceinline__ int lb_fast_accept(LbState* Sm, LbVecs<T>* Vl, const LbHist<T>& Hh, LbWork<T>& W, const LbOpts& O,
                                              double f_new, const T* gnew, int lane) {
    if (wave_uniform(Sm->phase) != PH_LS_FIRST) return -1;
    const double c1 = 1e-4, c2 = 0.9;
    const bool in = LB_EPL * lane < LB_D;
    const int i0 = in ? LB_EPL * lane : 0;
    LbVecs<T>& V = Vl[lane];                                  // (lanes past the vector hold zeros)
    T x[LB_EPL], d[LB_EPL], g[LB_EPL];
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) { x[e] = V.x[e]; d[e] = V.d[e]; g[e] = V.g[e]; }
    const double t = Sm->t, gtd = Sm->gtd, f0 = Sm->f0, prev_loss = Sm->prev_loss, d_norm = Sm->d_norm;
    const int n = Sm->n, cur = Sm->cur_evals + 1, hist_len0 = Sm->hist_len, head0 = Sm->hist_head;
    const double gtd_new = (double)vdot<T>(gnew, d);
    bool ok = !(f_new > (f0 + c1 * t * gtd)) && (fabs(gtd_new) <= -c2 * gtd);                  // :56, :63
    ok = ok && n != O.max_iter && cur < O.max_eval;                                                // :419-423
    ok = ok && !(vmaxabs(gnew) <= O.tol_grad);                                                     // :425
    ok = ok && !(d_norm * fabs(t) <= O.tol_change) && !(fabs(f_new - prev_loss) < O.tol_change);   // :428-434
    // the curvature pair of the next iteration (:318-335; g - prev_g with prev_g == the gradient the search started from)
    T y[LB_EPL], sv[LB_EPL];
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) { y[e] = gnew[e] - g[e]; sv[e] = d[e] * (T)t; }
    const T ys = vdot<T>(y, sv);
    const bool acc_pair = ys > (T)1e-10;
    int hist_len = hist_len0, head = head0;
    if (acc_pair) {
        if (hist_len == O.history) head = (head + 1) % LB_HIST; else hist_len += 1;
    }
    ok = ok && hist_len > 0;
    if (!wave_uniform((int)ok)) return -1;
    // ---- commit ----
    int slot = -1;
    double Hn = Sm->H;
    if (acc_pair) {
        slot = (head + hist_len - 1) % LB_HIST;
        if (LB_EPL * lane < Hh.ld) {
#pragma unroll
            for (int e = 0; e < LB_EPL; ++e) {
                Hh.dirs[slot * Hh.ld + LB_EPL * lane + e] = y[e];
                Hh.stps[slot * Hh.ld + LB_EPL * lane + e] = sv[e];
            }
        }
        if (lane == 0) { Hh.ro[slot] = (T)1 / ys; if (Hh.ys) Hh.ys[slot] = ys; }
        Hn = (double)(ys / vdot<T>(y, y));
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) {
        V.gprev[e] = g[e];                                                  // gprev = g                  (:46 of the search)
        V.bg0[e] = gnew[e];                                                 // bracket_g[0]
        V.g[e] = gnew[e];                                                   // flat_grad = g_new          (:397)
        V.x[e] = fma((T)t, d[e], x[e]);                                     // x += t d                   (:398)
        if (in) W.qv[i0 + e] = -gnew[e];
    }
    if (lane == 0) {
        Sm->n_closure += 1;
        Sm->ls_evals = 1; Sm->t_prev = 0.0; Sm->f_prev = f0; Sm->gtd_prev = gtd; Sm->ls_it = 0;
        Sm->br0 = t; Sm->bf0 = f_new; Sm->bgtd0 = gtd_new; Sm->nbr = 1; Sm->ls_done = 1;
        Sm->insuf = 0; Sm->low = 0; Sm->high = 1;
        Sm->loss = f_new; Sm->t = t; Sm->cur_evals = cur;
        Sm->n = n + 1; Sm->n_iter += 1; Sm->n_lbfgs += 1;
        Sm->ins_slot = slot; Sm->hist_len = hist_len; Sm->hist_head = head; Sm->H = Hn;
        Sm->phase = PH_DIRECTION;
        W.n = hist_len; W.head = head; W.ins_slot = slot; W.Hdiag = (T)Hn;
    }
    return 1;
}

// (2) behind the direction: the next search starts (lbfgs_ls.py:360-380, 43-53) unless the directional derivative says
//     stop.  Returns 0 (trial point emitted into xt) like lbfgs_advance.
template <typename T>
__device__ __forceinline__ int lb_fast_resume(LbState* Sm, LbVecs<T>* Vl, LbWork<T>& W, const LbOpts& O, T* xt, int lane) {
    if (wave_uniform(Sm->phase) != PH_DIRECTION) return -1;
    const bool in = LB_EPL * lane < LB_D;
    const int i0 = in ? LB_EPL * lane : 0;
    LbVecs<T>& V = Vl[lane];
    T x[LB_EPL], d[LB_EPL], g[LB_EPL];
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) { x[e] = V.x[e]; d[e] = in ? W.dv[i0 + e] : (T)0; g[e] = V.g[e]; }
    const double loss = Sm->loss, t = O.lr;                                  // (n_iter > 1 behind a direction: t = lr, :372-373)
    const double gtd = (double)vdot<T>(g, d);
    if (wave_uniform((int)(gtd > -O.tol_change))) return -1;               // :379-380 -> the general machine ends the step
    const double d_norm = vmaxabs(d);
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) { V.d[e] = d[e]; V.pg[e] = g[e]; }
    if (lane == 0) {
        Sm->prev_loss = loss; Sm->t = t; Sm->gtd = gtd; Sm->d_norm = d_norm; Sm->f0 = loss; Sm->ls_evals = 0;
        Sm->phase = PH_LS_FIRST;
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) xt[e] = fma((T)t, d[e], x[e]);
    return 0;
}

#ifdef MVFIT_LB_CHECK
static __device__ unsigned g_lb_check[4];      // [0] fast transitions checked, [1] mismatching words, [2] first mismatching word index + 1
template <typename T>
__device__ __forceinline__ void lb_check_compare(const LbState& Sa, const LbVecs<T>& Va, const LbState* Sm, const LbVecs<T>* Vl, int lane) {
    unsigned bad = 0, first = 0;
    const unsigned* a = reinterpret_cast<const unsigned*>(&Sa);
    const unsigned* b = reinterpret_cast<const unsigned*>(Sm);
    for (int i = 0; i < (int)(sizeof(LbState) / 4); ++i) {
        // pad words and the reuse flag the fast path does not touch are compared too: both start from the same image
        if (a[i] != b[i]) { bad += 1; if (!first) first = (unsigned)i + 1; }
    }
    const LbVecs<T> Vb = Vl[lane];
    unsigned vb = 0;
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) {
        vb += (Va.x[e] != Vb.x[e]) + (Va.d[e] != Vb.d[e]) + (Va.g[e] != Vb.g[e]) + (Va.pg[e] != Vb.pg[e]) +
              (Va.gprev[e] != Vb.gprev[e]) + (Va.bg0[e] != Vb.bg0[e]) + (Va.bg1[e] != Vb.bg1[e]);
    }
    vb = (unsigned)wave64_sum((float)vb);
    if (lane == 0) {
        atomicAdd(&g_lb_check[0], 1u);
        if (bad + vb) { atomicAdd(&g_lb_check[1], bad + vb); if (first) atomicMax(&g_lb_check[2], first); else atomicMax(&g_lb_check[2], 1000u); }
    }
}
#endif

// One optimiser round, shared by every kernel that drives the state machine (fit_step_kernel, fit_persistent_kernel,
// lbfgs_kat_kernel): consume (f_new, gnew) of the closure just evaluated; when the machine asks for a search
// direction, run `direction()` with the whole workgroup and resume WITH THE SAME (f_new, gnew) - the resumed call may
// leave through `gtd > -tolerance_change` (lbfgs_ls.py:379-380) into run_fitting's gtol test (fitting.py:115-116),
// which reads the .grad the last closure call left, i.e. this round's gnew.  Called by all NT threads (wave 0 = the
// optimiser wave; S, V, gnew, xt are meaningful in wave 0 only).  On return xt = the next trial point.
// REUSE: the instantiation for O.reuse_outer (MVFIT_F_REUSE_OUTER_VALUE); without it the round is the plain two-call form.
template <typename T, int NT, bool REUSE, typename DirFn>
__device__ __forceinline__ void lbfgs_round(LbState* Sm, LbVecs<T>* Vl, const LbHist<T>& Hh, LbWork<T>& W, const LbOpts& O,
                                            double f_new, const T* gnew, T* xt, int tid, double* stage_final,
                                            DirFn&& direction) {
    // one call of the state machine on the state in memory: the fast transitions first, else the general walk IN PLACE
    // (nothing of the state is live in registers across direction()).  which: 0 = a closure was evaluated, 1 = resumed
    // behind a direction.
    auto call = [&](int which, double fv, const T* gv, bool virt) -> int {
        const int lane = tid;
#ifdef MVFIT_LB_CHECK
        LbState S0 = *Sm;
        LbVecs<T> V0 = Vl[lane];
#endif
        int need = -1;
        const long long t_call = PH_CLK();
        const int ph_in = Sm->phase;
        (void)ph_in;
        if (which == 1) need = lb_fast_resume<T>(Sm, Vl, W, O, xt, lane);
        else if (!virt) need = lb_fast_accept<T>(Sm, Vl, Hh, W, O, fv, gv, lane);
        if (need >= 0) { PH_W(64 + 2 * which, 0, t_call); PH_ADD(65 + 2 * which, 1); }      // [64,65] fast accept, [66,67] fast resume: cycles, calls
#ifdef MVFIT_LB_CHECK
        if (need >= 0) {           // the general machine on the snapshot must arrive at the same words (it rewrites the same history row / work vectors)
            T xt2[LB_EPL];
            const int need2 = lbfgs_advance<T>(S0, V0, Hh, W, O, fv, gv, xt2, lane, stage_final, virt);
            lb_check_compare(S0, V0, Sm, Vl, lane);
            if (lane == 0 && need2 != need) atomicAdd(&g_lb_check[1], 1000000u);
            if (which == 1) { unsigned d_ = 0; for (int e = 0; e < LB_EPL; ++e) d_ += xt2[e] != xt[e]; if (d_) atomicAdd(&g_lb_check[1], 100000u); }
        }
#endif
        if (need < 0) {
            need = lbfgs_advance<T>(*Sm, Vl[lane], Hh, W, O, fv, gv, xt, lane, stage_final, virt);      // wave-uniform branch
            // general machine: cycles and calls by the phase it was entered in ([68,69] step start, first trial, bracket, zoom,
            // [76,77] behind a direction)
            const int slot = 68 + 2 * min(ph_in, 4);
            (void)slot;
            PH_W(slot, 0, t_call); PH_ADD(slot + 1, 1);
        }
        return need;
    };
    if constexpr (!REUSE) {
        // call; if it asks for a direction: direction, call again - written as a two-trip loop that is NOT unrolled, so that
        // the state machine is in the kernel once (~10 KB less code per round to stream through the instruction cache)
#pragma nounroll
        for (int trip = 0; trip < 2; ++trip) {
            if (tid < 64) {
                const int need = call(trip, f_new, gnew, false);
                if (tid == 0 && trip == 0) W.need_dir = need;
            }
            if (trip == 1) break;
            __syncthreads();
            if (!W.need_dir) break;                        // block-uniform
            direction();
        }
        return;
    }
    // Opt-in reuse (O.reuse_outer, a kernel argument: uniform): while the machine sits at a step start whose opening
    // closure call would return what it already holds, (loss, g) are fed back without an evaluation - further passes of
    // the same loop.  Bounded: every pass either emits a line-search trial point (a real closure follows) or ends a
    // step / stage.
    T gv[LB_EPL];
    double fv = f_new;
    if (tid < 64) {
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) gv[e] = gnew[e];
    }
    bool virt = false;
    for (int guard = 0; guard < 64; ++guard) {
        if (tid < 64) {
            const int need = call(0, fv, gv, virt);
            if (tid == 0) W.need_dir = need;
        }
        __syncthreads();
        if (W.need_dir) {                                  // block-uniform
            direction();
            if (tid < 64) call(1, fv, gv, virt);
        }
        __syncthreads();
        if (tid == 0) W.need_dir = (Sm->phase == PH_STEP_START && Sm->reuse_ok && !Sm->status) ? 1 : 0;
        __syncthreads();
        if (!W.need_dir) return;                           // block-uniform
        if (tid < 64) {
#pragma unroll
            for (int e = 0; e < LB_EPL; ++e) gv[e] = Vl[tid].g[e];
            fv = Sm->loss;
        }
        __syncthreads();                                   // (every lane has read the flag / the loss)
        if (tid == 0) Sm->reuse_ok = 0;
        virt = true;
        __syncthreads();                                   // W.need_dir is rewritten by the next pass
    }
}

}  // namespace mvfit
