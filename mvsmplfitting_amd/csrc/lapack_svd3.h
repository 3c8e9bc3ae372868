// 3 x 3 singular value decomposition along the path LAPACK's DGESDD takes for a matrix of this size, so that the SIGNS
// of the singular-vector pairs are the ones np.linalg.svd returns.
//
// Why: the reference's umeyama (code/utils/umeyama.py:51,73) forms U diag(d) V.T with V already numpy's Vh.  That
// product is not invariant under the sign freedom of the factorisation ((u_k, v_k) -> (-u_k, -v_k)), so its value is
// "whatever LAPACK returned"; to reproduce the reference's initial rotation / translation the device has to return
// the same pairs.  Netlib LAPACK 3.10+ (the one inside NumPy's OpenBLAS), M = N = 3 < MNTHR, JOBZ = 'A' (path 5):
//     DGEBRD -> DGEBD2   Householder bidiagonalisation (DLARFG: beta = -sign(alpha) * norm; DLARF)
//     DBDSDC('U','I')    N <= SMLSIZ -> DLASDQ -> DBDSQR: implicit zero-shift / shifted QR sweeps built from DLARTG
//                        rotations (c >= 0, r with the sign of f), DLASV2 for 2 x 2 blocks, negative singular values
//                        flipped together with their VT row, selection sort into descending order
//     DORMBR('Q','L','N')  U  := H(1) H(2) H(3) U        DORMBR('P','R','T')  VT(:, 2:3) := VT(:, 2:3) G(2) G(1)
// oracle/lapack_svd3_np.py is the same text in Python; tests/test_umeyama.py compiles THIS header for the host and
// checks both against np.linalg.svd (signs exact, values 1e-12) on random, graded, triangular and rank-deficient inputs.
// Plain scalar double code, one thread per matrix; no HIP-specific constructs (MVFIT_HD is empty for the host build).
#pragma once
#include <math.h>

#ifndef MVFIT_HD
#ifdef __HIPCC__
#define MVFIT_HD __host__ __device__
#else
#define MVFIT_HD
#endif
#endif

// no fused multiply-adds here: the rounding of every operation is LAPACK's (a sign decision can hang on one)
#pragma clang fp contract(off)

namespace mvfit {
namespace lapack3 {

constexpr double kEps = 1.1102230246251565e-16;       // DLAMCH('Epsilon') = 2^-53
constexpr double kSafmin = 2.2250738585072014e-308;   // DLAMCH('Safe minimum')

MVFIT_HD inline double sgn(double a, double b) { return copysign(fabs(a), b); }      // Fortran SIGN(a, b)

// LAPACK 3.10+ DLARTG: c >= 0, r carries the sign of f
MVFIT_HD inline void dlartg(double f, double g, double& c, double& s, double& r) {
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; return; }
    if (f == 0.0) { c = 0.0; s = sgn(1.0, g); r = fabs(g); return; }
    const double d = sqrt(f * f + g * g);             // (the scaled branch only matters near over / underflow)
    c = fabs(f) / d;
    r = sgn(d, f);
    s = g / r;
}

MVFIT_HD inline double dlas2_min(double f, double g, double h) {                   // smaller singular value of [[f, g], [0, h]]
    const double fa = fabs(f), ga = fabs(g), ha = fabs(h);
    const double fhmn = fmin(fa, ha), fhmx = fmax(fa, ha);
    if (fhmn == 0.0) return 0.0;
    if (ga < fhmx) {
        const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx, au = (ga / fhmx) * (ga / fhmx);
        const double c = 2.0 / (sqrt(as * as + au) + sqrt(at * at + au));
        return fhmn * c;
    }
    const double au = fhmx / ga;
    if (au == 0.0) return (fhmn * fhmx) / ga;
    const double as = 1.0 + fhmn / fhmx, at = (fhmx - fhmn) / fhmx;
    const double c = 1.0 / (sqrt(1.0 + (as * au) * (as * au)) + sqrt(1.0 + (at * au) * (at * au)));
    const double ssmin = (fhmn * c) * au;
    return ssmin + ssmin;
}

// SVD of [[f, g], [0, h]]
MVFIT_HD inline void dlasv2(double f, double g, double h, double& ssmin, double& ssmax, double& snr, double& csr, double& snl,
                            double& csl) {
    double ft = f, fa = fabs(f), ht = h, ha = fabs(h);
    int pmax = 1;
    const bool swap = ha > fa;
    if (swap) {
        pmax = 3;
        double t = ft; ft = ht; ht = t;
        t = fa; fa = ha; ha = t;
    }
    const double gt = g, ga = fabs(g);
    double clt, crt, slt, srt;
    if (ga == 0.0) {
        ssmin = ha; ssmax = fa; clt = 1.0; crt = 1.0; slt = 0.0; srt = 0.0;
    } else {
        bool gasmal = true;
        if (ga > fa) {
            pmax = 2;
            if (fa / ga < kEps) {
                gasmal = false;
                ssmax = ga;
                ssmin = ha > 1.0 ? fa / (ga / ha) : (fa / ga) * ha;
                clt = 1.0; slt = ht / gt; srt = 1.0; crt = ft / gt;
            }
        }
        if (gasmal) {
            const double d = fa - ha;
            double l = (d == fa) ? 1.0 : d / fa;
            const double m = gt / ft;
            double t = 2.0 - l;
            const double mm = m * m, tt = t * t;
            const double s = sqrt(tt + mm);
            const double r = (l == 0.0) ? fabs(m) : sqrt(l * l + mm);
            const double a = 0.5 * (s + r);
            ssmin = ha / a;
            ssmax = fa * a;
            if (mm == 0.0) {
                if (l == 0.0) t = sgn(2.0, ft) * sgn(1.0, gt);
                else t = gt / sgn(d, ft) + m / t;
            } else {
                t = (m / (s + t) + m / (r + l)) * (1.0 + a);
            }
            l = sqrt(t * t + 4.0);
            crt = 2.0 / l;
            srt = t / l;
            clt = (crt + srt * m) / a;
            slt = (ht / ft) * srt / a;
        }
    }
    if (swap) { csl = srt; snl = crt; csr = slt; snr = clt; }
    else { csl = clt; snl = slt; csr = crt; snr = srt; }
    double tsign;
    if (pmax == 1) tsign = sgn(1.0, csr) * sgn(1.0, csl) * sgn(1.0, f);
    else if (pmax == 2) tsign = sgn(1.0, snr) * sgn(1.0, csl) * sgn(1.0, g);
    else tsign = sgn(1.0, snr) * sgn(1.0, snl) * sgn(1.0, h);
    ssmax = sgn(ssmax, tsign);
    ssmin = sgn(ssmin, tsign * sgn(1.0, f) * sgn(1.0, h));
}

// DLARFG on (alpha, x[0 .. nx)): H = I - tau [1; v] [1; v]^T maps [alpha; x] to [beta; 0]; x is overwritten with v
MVFIT_HD inline void dlarfg(double& alpha, double* x, int nx, double& tau) {
    if (nx == 0) { tau = 0.0; return; }
    double xnorm;
    if (nx == 1) xnorm = fabs(x[0]);
    else {                                              // dnrm2 (scaled form is irrelevant away from over / underflow)
        double ss = 0.0;
        for (int i = 0; i < nx; ++i) ss += x[i] * x[i];
        xnorm = sqrt(ss);
    }
    if (xnorm == 0.0) { tau = 0.0; return; }
    const double beta = -sgn(hypot(alpha, xnorm), alpha);      // dlapy2
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = 0; i < nx; ++i) x[i] *= sc;
    alpha = beta;
}

// DLASR('L', 'V', dir) on rows ll + j, ll + j + 1 of VT / DLASR('R', 'V', dir) on columns of U (row-major 3 x 3)
MVFIT_HD inline void lasr_rows(double* VT, int ll, const double* cs, const double* sn, int cnt, bool fwd) {
    for (int q = 0; q < cnt; ++q) {
        const int j = fwd ? q : cnt - 1 - q;
        for (int k = 0; k < 3; ++k) {
            const double temp = VT[3 * (ll + j + 1) + k];
            VT[3 * (ll + j + 1) + k] = cs[j] * temp - sn[j] * VT[3 * (ll + j) + k];
            VT[3 * (ll + j) + k] = sn[j] * temp + cs[j] * VT[3 * (ll + j) + k];
        }
    }
}
MVFIT_HD inline void lasr_cols(double* U, int ll, const double* cs, const double* sn, int cnt, bool fwd) {
    for (int q = 0; q < cnt; ++q) {
        const int j = fwd ? q : cnt - 1 - q;
        for (int k = 0; k < 3; ++k) {
            const double temp = U[3 * k + ll + j + 1];
            U[3 * k + ll + j + 1] = cs[j] * temp - sn[j] * U[3 * k + ll + j];
            U[3 * k + ll + j] = sn[j] * temp + cs[j] * U[3 * k + ll + j];
        }
    }
}

// DBDSQR('U', 3, ncvt = 3, nru = 3, ncc = 0): d[3], e[2]; VT, U (row-major) start as the identity
MVFIT_HD inline void dbdsqr3(double* d, double* e, double* VT, double* U) {
    const int n = 3, maxitr = 6;
    const double tolmul = fmax(10.0, fmin(100.0, pow(kEps, -0.125)));
    const double tol = tolmul * kEps;
    double sminoa = fabs(d[0]);
    if (sminoa != 0.0) {
        double mu = sminoa;
        for (int i = 1; i < n; ++i) {
            mu = fabs(d[i]) * (mu / (mu + fabs(e[i - 1])));
            sminoa = fmin(sminoa, mu);
            if (sminoa == 0.0) break;
        }
    }
    sminoa = sminoa / sqrt((double)n);
    const double thresh = fmax(tol * sminoa, maxitr * ((double)n * ((double)n * kSafmin)));
    const int maxitdivn = maxitr * n;
    int iterdivn = 0, it = -1, oldll = -1, oldm = -1, idir = 0;
    int m = n;                                          // 1-based index of the last unconverged element, as in the Fortran
    for (;;) {
        if (m <= 1) break;
        if (it >= n) {
            it -= n;
            if (++iterdivn >= maxitdivn) break;         // (no convergence: never observed for 3 x 3)
        }
        double smax = fabs(d[m - 1]);
        bool split = false;
        int ll = 0;
        for (int lll = 1; lll < m; ++lll) {
            ll = m - lll;
            const double abss = fabs(d[ll - 1]), abse = fabs(e[ll - 1]);
            if (abse <= thresh) { split = true; break; }
            smax = fmax(smax, fmax(abss, abse));
        }
        if (split) {
            e[ll - 1] = 0.0;
            if (ll == m - 1) { m -= 1; continue; }
        } else {
            ll = 0;
        }
        ll += 1;
        if (ll == m - 1) {                              // 2 x 2 block
            double sigmn, sigmx, sinr, cosr, sinl, cosl;
            dlasv2(d[m - 2], e[m - 2], d[m - 1], sigmn, sigmx, sinr, cosr, sinl, cosl);
            d[m - 2] = sigmx; e[m - 2] = 0.0; d[m - 1] = sigmn;
            for (int k = 0; k < 3; ++k) {               // DROT on rows m-1, m of VT and columns m-1, m of U
                const double x = VT[3 * (m - 2) + k], y = VT[3 * (m - 1) + k];
                VT[3 * (m - 2) + k] = cosr * x + sinr * y;
                VT[3 * (m - 1) + k] = cosr * y - sinr * x;
                const double p = U[3 * k + m - 2], q = U[3 * k + m - 1];
                U[3 * k + m - 2] = cosl * p + sinl * q;
                U[3 * k + m - 1] = cosl * q - sinl * p;
            }
            m -= 2;
            continue;
        }
        if (ll > oldm || m < oldll) idir = fabs(d[ll - 1]) >= fabs(d[m - 1]) ? 1 : 2;
        double sminl = 0.0;
        bool conv = false;
        if (idir == 1) {
            if (fabs(e[m - 2]) <= fabs(tol) * fabs(d[m - 1])) { e[m - 2] = 0.0; continue; }
            double mu = fabs(d[ll - 1]);
            sminl = mu;
            for (int lll = ll; lll < m; ++lll) {
                if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = true; break; }
                mu = fabs(d[lll]) * (mu / (mu + fabs(e[lll - 1])));
                sminl = fmin(sminl, mu);
            }
        } else {
            if (fabs(e[ll - 1]) <= fabs(tol) * fabs(d[ll - 1])) { e[ll - 1] = 0.0; continue; }
            double mu = fabs(d[m - 1]);
            sminl = mu;
            for (int lll = m - 1; lll >= ll; --lll) {
                if (fabs(e[lll - 1]) <= tol * mu) { e[lll - 1] = 0.0; conv = true; break; }
                mu = fabs(d[lll - 1]) * (mu / (mu + fabs(e[lll - 1])));
                sminl = fmin(sminl, mu);
            }
        }
        if (conv) continue;
        oldll = ll; oldm = m;
        double shift;
        if (n * tol * (sminl / smax) <= fmax(kEps, 0.01 * tol)) {
            shift = 0.0;
        } else {
            double sll;
            if (idir == 1) { sll = fabs(d[ll - 1]); shift = dlas2_min(d[m - 2], e[m - 2], d[m - 1]); }
            else { sll = fabs(d[m - 1]); shift = dlas2_min(d[ll - 1], e[ll - 1], d[ll]); }
            if (sll > 0.0 && (shift / sll) * (shift / sll) < kEps) shift = 0.0;
        }
        it += m - ll;
        const int cnt = m - ll;                         // 2 here (the whole 3 x 3 block)
        double wc[2], ws[2], woc[2], wos[2];
        if (shift == 0.0) {
            if (idir == 1) {
                double cs = 1.0, sn = 0.0, oldcs = 1.0, oldsn = 0.0, r;
                for (int i = ll; i < m; ++i) {
                    dlartg(d[i - 1] * cs, e[i - 1], cs, sn, r);
                    if (i > ll) e[i - 2] = oldsn * r;
                    dlartg(oldcs * r, d[i] * sn, oldcs, oldsn, d[i - 1]);
                    wc[i - ll] = cs; ws[i - ll] = sn; woc[i - ll] = oldcs; wos[i - ll] = oldsn;
                }
                const double h = d[m - 1] * cs;
                d[m - 1] = h * oldcs;
                e[m - 2] = h * oldsn;
                lasr_rows(VT, ll - 1, wc, ws, cnt, true);
                lasr_cols(U, ll - 1, woc, wos, cnt, true);
                if (fabs(e[m - 2]) <= thresh) e[m - 2] = 0.0;
            } else {
                double cs = 1.0, sn = 0.0, oldcs = 1.0, oldsn = 0.0, r;
                for (int i = m; i > ll; --i) {
                    dlartg(d[i - 1] * cs, e[i - 2], cs, sn, r);
                    if (i < m) e[i - 1] = oldsn * r;
                    dlartg(oldcs * r, d[i - 2] * sn, oldcs, oldsn, d[i - 1]);
                    wc[i - ll - 1] = cs; ws[i - ll - 1] = -sn; woc[i - ll - 1] = oldcs; wos[i - ll - 1] = -oldsn;
                }
                const double h = d[ll - 1] * cs;
                d[ll - 1] = h * oldcs;
                e[ll - 1] = h * oldsn;
                lasr_rows(VT, ll - 1, woc, wos, cnt, false);
                lasr_cols(U, ll - 1, wc, ws, cnt, false);
                if (fabs(e[ll - 1]) <= thresh) e[ll - 1] = 0.0;
            }
        } else {
            if (idir == 1) {
                double f = (fabs(d[ll - 1]) - shift) * (sgn(1.0, d[ll - 1]) + shift / d[ll - 1]);
                double g = e[ll - 1];
                for (int i = ll; i < m; ++i) {
                    double cosr, sinr, cosl, sinl, r;
                    dlartg(f, g, cosr, sinr, r);
                    if (i > ll) e[i - 2] = r;
                    f = cosr * d[i - 1] + sinr * e[i - 1];
                    e[i - 1] = cosr * e[i - 1] - sinr * d[i - 1];
                    g = sinr * d[i];
                    d[i] = cosr * d[i];
                    dlartg(f, g, cosl, sinl, r);
                    d[i - 1] = r;
                    f = cosl * e[i - 1] + sinl * d[i];
                    d[i] = cosl * d[i] - sinl * e[i - 1];
                    if (i < m - 1) { g = sinl * e[i]; e[i] = cosl * e[i]; }
                    wc[i - ll] = cosr; ws[i - ll] = sinr; woc[i - ll] = cosl; wos[i - ll] = sinl;
                }
                e[m - 2] = f;
                lasr_rows(VT, ll - 1, wc, ws, cnt, true);
                lasr_cols(U, ll - 1, woc, wos, cnt, true);
                if (fabs(e[m - 2]) <= thresh) e[m - 2] = 0.0;
            } else {
                double f = (fabs(d[m - 1]) - shift) * (sgn(1.0, d[m - 1]) + shift / d[m - 1]);
                double g = e[m - 2];
                for (int i = m; i > ll; --i) {
                    double cosr, sinr, cosl, sinl, r;
                    dlartg(f, g, cosr, sinr, r);
                    if (i < m) e[i - 1] = r;
                    f = cosr * d[i - 1] + sinr * e[i - 2];
                    e[i - 2] = cosr * e[i - 2] - sinr * d[i - 1];
                    g = sinr * d[i - 2];
                    d[i - 2] = cosr * d[i - 2];
                    dlartg(f, g, cosl, sinl, r);
                    d[i - 1] = r;
                    f = cosl * e[i - 2] + sinl * d[i - 2];
                    d[i - 2] = cosl * d[i - 2] - sinl * e[i - 2];
                    if (i > ll + 1) { g = sinl * e[i - 3]; e[i - 3] = cosl * e[i - 3]; }
                    wc[i - ll - 1] = cosr; ws[i - ll - 1] = -sinr; woc[i - ll - 1] = cosl; wos[i - ll - 1] = -sinl;
                }
                e[ll - 1] = f;
                if (fabs(e[ll - 1]) <= thresh) e[ll - 1] = 0.0;
                lasr_rows(VT, ll - 1, woc, wos, cnt, false);
                lasr_cols(U, ll - 1, wc, ws, cnt, false);
            }
        }
    }
    // make the singular values positive (a negative one takes its VT row along)
    for (int i = 0; i < n; ++i) {
        if (d[i] == 0.0) d[i] = 0.0;                    // "avoid -ZERO": no flip for a negative zero
        if (d[i] < 0.0) {
            d[i] = -d[i];
            for (int k = 0; k < 3; ++k) VT[3 * i + k] = -VT[3 * i + k];
        }
    }
    // selection sort into decreasing order, one transposition per singular vector
    for (int i = 1; i < n; ++i) {
        int isub = 1;
        double smin = d[0];
        for (int j = 2; j <= n + 1 - i; ++j)
            if (d[j - 1] <= smin) { isub = j; smin = d[j - 1]; }
        const int last = n + 1 - i;
        if (isub != last) {
            d[isub - 1] = d[last - 1];
            d[last - 1] = smin;
            for (int k = 0; k < 3; ++k) {
                double t = VT[3 * (isub - 1) + k]; VT[3 * (isub - 1) + k] = VT[3 * (last - 1) + k]; VT[3 * (last - 1) + k] = t;
                t = U[3 * k + isub - 1]; U[3 * k + isub - 1] = U[3 * k + last - 1]; U[3 * k + last - 1] = t;
            }
        }
    }
}

// A (row-major 3 x 3) = U diag(S) Vh with S descending - the factors np.linalg.svd(A) returns
MVFIT_HD inline void svd3(const double* A, double* U, double* S, double* Vh) {
    const int n = 3;
    double a[9];
    for (int i = 0; i < 9; ++i) a[i] = A[i];
    double d[3], e[2], tauq[3], taup[3];
    // ---- DGEBD2: a(i+1:, i) keeps the Householder vector of H(i), a(i, i+2:) the one of G(i) (implicit leading 1) ----
    for (int i = 0; i < n; ++i) {
        double col[2];
        const int nx = n - 1 - i;
        for (int r = 0; r < nx; ++r) col[r] = a[3 * (i + 1 + r) + i];
        double alpha = a[3 * i + i];
        dlarfg(alpha, col, nx, tauq[i]);
        d[i] = alpha;
        for (int r = 0; r < nx; ++r) a[3 * (i + 1 + r) + i] = col[r];
        if (i < n - 1 && tauq[i] != 0.0) {             // DLARF('Left') on a(i:, i+1:)
            for (int c = i + 1; c < n; ++c) {
                double w = a[3 * i + c];
                for (int r = i + 1; r < n; ++r) w += a[3 * r + i] * a[3 * r + c];
                a[3 * i + c] -= tauq[i] * w;
                for (int r = i + 1; r < n; ++r) a[3 * r + c] -= tauq[i] * a[3 * r + i] * w;
            }
        }
        if (i < n - 1) {
            double row[1];
            const int ny = n - 2 - i;
            for (int c = 0; c < ny; ++c) row[c] = a[3 * i + i + 2 + c];
            double al2 = a[3 * i + i + 1];
            dlarfg(al2, row, ny, taup[i]);
            e[i] = al2;
            for (int c = 0; c < ny; ++c) a[3 * i + i + 2 + c] = row[c];
            if (taup[i] != 0.0) {                       // DLARF('Right') on a(i+1:, i+1:)
                for (int r = i + 1; r < n; ++r) {
                    double w = a[3 * r + i + 1];
                    for (int c = i + 2; c < n; ++c) w += a[3 * r + c] * a[3 * i + c];
                    a[3 * r + i + 1] -= taup[i] * w;
                    for (int c = i + 2; c < n; ++c) a[3 * r + c] -= taup[i] * w * a[3 * i + c];
                }
            }
        } else {
            taup[i] = 0.0;
        }
    }
    // ---- DBDSDC('U','I') -> DLASDQ -> DBDSQR ----
    for (int i = 0; i < 9; ++i) { U[i] = (i % 4 == 0) ? 1.0 : 0.0; Vh[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    dbdsqr3(d, e, Vh, U);
    // ---- DORMBR('Q','L','N'): U := H(1) H(2) H(3) U (H(3) first) ----
    for (int i = n - 1; i >= 0; --i) {
        if (tauq[i] == 0.0) continue;
        for (int c = 0; c < n; ++c) {
            double w = U[3 * i + c];
            for (int r = i + 1; r < n; ++r) w += a[3 * r + i] * U[3 * r + c];
            U[3 * i + c] -= tauq[i] * w;
            for (int r = i + 1; r < n; ++r) U[3 * r + c] -= tauq[i] * a[3 * r + i] * w;
        }
    }
    // ---- DORMBR('P','R','T'): VT(:, 2:3) := VT(:, 2:3) G(2) G(1) ----
    for (int i = n - 2; i >= 0; --i) {
        if (taup[i] == 0.0) continue;
        for (int r = 0; r < n; ++r) {
            double w = Vh[3 * r + i + 1];
            for (int c = i + 2; c < n; ++c) w += Vh[3 * r + c] * a[3 * i + c];
            Vh[3 * r + i + 1] -= taup[i] * w;
            for (int c = i + 2; c < n; ++c) Vh[3 * r + c] -= taup[i] * w * a[3 * i + c];
        }
    }
    for (int i = 0; i < 3; ++i) S[i] = d[i];
}

}  // namespace lapack3
}  // namespace mvfit
