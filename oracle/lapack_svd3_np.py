"""TEST INFRASTRUCTURE - the path LAPACK's DGESDD takes for a 3 x 3 matrix, restated in plain Python floats
(`np.linalg.svd(A)` -> LAPACK dgesdd, JOBZ = 'A'), so that the SIGNS of the singular-vector pairs - a free choice of
the factorisation that the reference's umeyama is sensitive to (code/utils/umeyama.py:51,73: U diag(d) V.T with V
already Vh) - can be reproduced on the device (csrc/init_guess.hip: svd3_lapack is the transcription of this file).

Netlib LAPACK (3.10+, the one inside NumPy's OpenBLAS), M = N = 3 < MNTHR: path 5, JOBZ = 'A':
    DGEBRD (-> DGEBD2: Householder bidiagonalisation, DLARFG with beta = -sign(alpha) * norm, DLARF)
    DBDSDC('U', 'I')  N <= SMLSIZ -> DLASDQ -> DBDSQR (implicit zero-shift / shifted QR sweeps with DLARTG rotations,
                      DLASV2 for 2 x 2 blocks, negative singular values flipped with their VT row, sort descending)
    DORMBR('Q', 'L', 'N')  U  := Q U      Q = H(1) H(2) H(3)
    DORMBR('P', 'R', 'T')  VT := VT P^T   P = G(1) G(2)
Pinned by tests/test_umeyama.py against np.linalg.svd on random / degenerate matrices (values to 1e-13, signs exact).
Never imported by the shipped package."""
from __future__ import annotations

import math

EPS = 2.0 ** -53                   # DLAMCH('Epsilon')
SAFMIN = 2.2250738585072014e-308   # DLAMCH('Safe minimum')


def _sign(a, b):
    """Fortran SIGN(a, b): |a| with the sign of b (b = +0 counts as positive, -0 as negative)."""
    return math.copysign(abs(a), b)


def dlartg(f, g):
    """LAPACK 3.10+ DLARTG (la_lartg.f90): c >= 0, r carries the sign of f.  -> (c, s, r)"""
    if g == 0.0:
        return 1.0, 0.0, f
    if f == 0.0:
        return 0.0, _sign(1.0, g), abs(g)
    d = math.sqrt(f * f + g * g)         # (the scaled branch only matters near over / underflow)
    c = abs(f) / d
    r = _sign(d, f)
    return c, g / r, r


def dlas2(f, g, h):
    fa, ga, ha = abs(f), abs(g), abs(h)
    fhmn, fhmx = min(fa, ha), max(fa, ha)
    if fhmn == 0.0:
        if fhmx == 0.0:
            return 0.0, ga
        return 0.0, max(fhmx, ga) * math.sqrt(1.0 + (min(fhmx, ga) / max(fhmx, ga)) ** 2)
    if ga < fhmx:
        as_ = 1.0 + fhmn / fhmx
        at = (fhmx - fhmn) / fhmx
        au = (ga / fhmx) ** 2
        c = 2.0 / (math.sqrt(as_ * as_ + au) + math.sqrt(at * at + au))
        return fhmn * c, fhmx / c
    au = fhmx / ga
    if au == 0.0:
        return (fhmn * fhmx) / ga, ga
    as_ = 1.0 + fhmn / fhmx
    at = (fhmx - fhmn) / fhmx
    c = 1.0 / (math.sqrt(1.0 + (as_ * au) ** 2) + math.sqrt(1.0 + (at * au) ** 2))
    ssmin = (fhmn * c) * au
    return ssmin + ssmin, ga / (c + c)


def dlasv2(f, g, h):
    """SVD of [[f, g], [0, h]] -> (ssmin, ssmax, snr, csr, snl, csl)."""
    ft, fa, ht, ha = f, abs(f), h, abs(h)
    pmax = 1
    swap = ha > fa
    if swap:
        pmax = 3
        ft, ht = ht, ft
        fa, ha = ha, fa
    gt, ga = g, abs(g)
    if ga == 0.0:
        ssmin, ssmax, clt, crt, slt, srt = ha, fa, 1.0, 1.0, 0.0, 0.0
    else:
        gasmal = True
        if ga > fa:
            pmax = 2
            if fa / ga < EPS:
                gasmal = False
                ssmax = ga
                ssmin = fa / (ga / ha) if ha > 1.0 else (fa / ga) * ha
                clt, slt, srt, crt = 1.0, ht / gt, 1.0, ft / gt
        if gasmal:
            d = fa - ha
            l = 1.0 if d == fa else d / fa
            m = gt / ft
            t = 2.0 - l
            mm, tt = m * m, t * t
            s = math.sqrt(tt + mm)
            r = abs(m) if l == 0.0 else math.sqrt(l * l + mm)
            a = 0.5 * (s + r)
            ssmin, ssmax = ha / a, fa * a
            if mm == 0.0:
                if l == 0.0:
                    t = _sign(2.0, ft) * _sign(1.0, gt)
                else:
                    t = gt / _sign(d, ft) + m / t
            else:
                t = (m / (s + t) + m / (r + l)) * (1.0 + a)
            l = math.sqrt(t * t + 4.0)
            crt, srt = 2.0 / l, t / l
            clt = (crt + srt * m) / a
            slt = (ht / ft) * srt / a
    if swap:
        csl, snl, csr, snr = srt, crt, slt, clt
    else:
        csl, snl, csr, snr = clt, slt, crt, srt
    if pmax == 1:
        tsign = _sign(1.0, csr) * _sign(1.0, csl) * _sign(1.0, f)
    elif pmax == 2:
        tsign = _sign(1.0, snr) * _sign(1.0, csl) * _sign(1.0, g)
    else:
        tsign = _sign(1.0, snr) * _sign(1.0, snl) * _sign(1.0, h)
    ssmax = _sign(ssmax, tsign)
    ssmin = _sign(ssmin, tsign * _sign(1.0, f) * _sign(1.0, h))
    return ssmin, ssmax, snr, csr, snl, csl


def dlarfg(alpha, x):
    """-> (beta, tau, v): H = I - tau [1; v] [1; v]^T maps [alpha; x] to [beta; 0]."""
    if len(x) == 0:
        return alpha, 0.0, []
    xnorm = math.sqrt(sum(t * t for t in x)) if len(x) > 1 else abs(x[0])      # dnrm2
    if xnorm == 0.0:
        return alpha, 0.0, list(x)
    beta = -_sign(math.hypot(alpha, xnorm), alpha)                               # dlapy2
    tau = (beta - alpha) / beta
    sc = 1.0 / (alpha - beta)
    return beta, tau, [t * sc for t in x]


def _rot_rows(M, i, j, c, s, n):
    """DROT on rows i, j: x' = c x + s y ; y' = c y - s x"""
    for k in range(n):
        x, y = M[i][k], M[j][k]
        M[i][k] = c * x + s * y
        M[j][k] = c * y - s * x


def _rot_cols(M, i, j, c, s, n):
    for k in range(n):
        x, y = M[k][i], M[k][j]
        M[k][i] = c * x + s * y
        M[k][j] = c * y - s * x


def _dlasr_left(VT, ll, cs, sn, count, forward, n):
    """DLASR('L', 'V', dir): rows ll + j, ll + j + 1 of VT for the count rotations (cs[j], sn[j])."""
    order = range(count) if forward else range(count - 1, -1, -1)
    for j in order:
        c, s = cs[j], sn[j]
        for k in range(n):
            temp = VT[ll + j + 1][k]
            VT[ll + j + 1][k] = c * temp - s * VT[ll + j][k]
            VT[ll + j][k] = s * temp + c * VT[ll + j][k]


def _dlasr_right(U, ll, cs, sn, count, forward, n):
    order = range(count) if forward else range(count - 1, -1, -1)
    for j in order:
        c, s = cs[j], sn[j]
        for k in range(n):
            temp = U[k][ll + j + 1]
            U[k][ll + j + 1] = c * temp - s * U[k][ll + j]
            U[k][ll + j] = s * temp + c * U[k][ll + j]


def dbdsqr(d, e, VT, U, n=3):
    """DBDSQR('U', n, ncvt = n, nru = n, ncc = 0): d [n], e [n - 1] upper bidiagonal; VT, U updated in place."""
    maxitr = 6
    tolmul = max(10.0, min(100.0, EPS ** -0.125))
    tol = tolmul * EPS
    smax = max([abs(v) for v in d] + [abs(v) for v in e])
    sminoa = abs(d[0])
    if sminoa != 0.0:
        mu = sminoa
        for i in range(1, n):
            mu = abs(d[i]) * (mu / (mu + abs(e[i - 1])))
            sminoa = min(sminoa, mu)
            if sminoa == 0.0:
                break
    sminoa = sminoa / math.sqrt(float(n))
    thresh = max(tol * sminoa, maxitr * (float(n) * (float(n) * SAFMIN)))
    maxitdivn = maxitr * n
    iterdivn = 0
    it = -1
    oldll, oldm = -1, -1
    idir = 0
    m = n                                  # 1-based index of the last unconverged element, as in the Fortran
    while True:
        if m <= 1:
            break
        if it >= n:
            it -= n
            iterdivn += 1
            if iterdivn >= maxitdivn:
                raise RuntimeError('dbdsqr: no convergence')
        # find the diagonal block to work on
        smax = abs(d[m - 1])
        split = False
        ll = 0
        for lll in range(1, m):
            ll = m - lll
            abss, abse = abs(d[ll - 1]), abs(e[ll - 1])
            if abse <= thresh:
                split = True
                break
            smax = max(smax, abss, abse)
        if split:
            e[ll - 1] = 0.0
            if ll == m - 1:
                m -= 1
                continue
        else:
            ll = 0
        ll += 1
        if ll == m - 1:
            sigmn, sigmx, sinr, cosr, sinl, cosl = dlasv2(d[m - 2], e[m - 2], d[m - 1])
            d[m - 2], e[m - 2], d[m - 1] = sigmx, 0.0, sigmn
            _rot_rows(VT, m - 2, m - 1, cosr, sinr, n)
            _rot_cols(U, m - 2, m - 1, cosl, sinl, n)
            m -= 2
            continue
        if ll > oldm or m < oldll:
            idir = 1 if abs(d[ll - 1]) >= abs(d[m - 1]) else 2
        # convergence tests
        if idir == 1:
            if abs(e[m - 2]) <= abs(tol) * abs(d[m - 1]):
                e[m - 2] = 0.0
                continue
            mu = abs(d[ll - 1])
            sminl = mu
            conv = False
            for lll in range(ll, m):
                if abs(e[lll - 1]) <= tol * mu:
                    e[lll - 1] = 0.0
                    conv = True
                    break
                mu = abs(d[lll]) * (mu / (mu + abs(e[lll - 1])))
                sminl = min(sminl, mu)
            if conv:
                continue
        else:
            if abs(e[ll - 1]) <= abs(tol) * abs(d[ll - 1]):
                e[ll - 1] = 0.0
                continue
            mu = abs(d[m - 1])
            sminl = mu
            conv = False
            for lll in range(m - 1, ll - 1, -1):
                if abs(e[lll - 1]) <= tol * mu:
                    e[lll - 1] = 0.0
                    conv = True
                    break
                mu = abs(d[lll - 1]) * (mu / (mu + abs(e[lll - 1])))
                sminl = min(sminl, mu)
            if conv:
                continue
        oldll, oldm = ll, m
        # shift
        if n * tol * (sminl / smax) <= max(EPS, 0.01 * tol):
            shift = 0.0
        else:
            if idir == 1:
                sll = abs(d[ll - 1])
                shift, _r = dlas2(d[m - 2], e[m - 2], d[m - 1])
            else:
                sll = abs(d[m - 1])
                shift, _r = dlas2(d[ll - 1], e[ll - 1], d[ll])
            if sll > 0.0 and (shift / sll) ** 2 < EPS:
                shift = 0.0
        it += m - ll
        cnt = m - ll
        w_c, w_s, w_oc, w_os = [0.0] * cnt, [0.0] * cnt, [0.0] * cnt, [0.0] * cnt
        if shift == 0.0:
            if idir == 1:
                cs, oldcs, oldsn = 1.0, 1.0, 0.0
                for i in range(ll, m):
                    cs, sn, r = dlartg(d[i - 1] * cs, e[i - 1])
                    if i > ll:
                        e[i - 2] = oldsn * r
                    oldcs, oldsn, d[i - 1] = dlartg(oldcs * r, d[i] * sn)
                    w_c[i - ll], w_s[i - ll], w_oc[i - ll], w_os[i - ll] = cs, sn, oldcs, oldsn
                h = d[m - 1] * cs
                d[m - 1] = h * oldcs
                e[m - 2] = h * oldsn
                _dlasr_left(VT, ll - 1, w_c, w_s, cnt, True, n)
                _dlasr_right(U, ll - 1, w_oc, w_os, cnt, True, n)
                if abs(e[m - 2]) <= thresh:
                    e[m - 2] = 0.0
            else:
                cs, oldcs, oldsn = 1.0, 1.0, 0.0
                for i in range(m, ll, -1):
                    cs, sn, r = dlartg(d[i - 1] * cs, e[i - 2])
                    if i < m:
                        e[i - 1] = oldsn * r
                    oldcs, oldsn, d[i - 1] = dlartg(oldcs * r, d[i - 2] * sn)
                    w_c[i - ll - 1], w_s[i - ll - 1], w_oc[i - ll - 1], w_os[i - ll - 1] = cs, -sn, oldcs, -oldsn
                h = d[ll - 1] * cs
                d[ll - 1] = h * oldcs
                e[ll - 1] = h * oldsn
                _dlasr_left(VT, ll - 1, w_oc, w_os, cnt, False, n)
                _dlasr_right(U, ll - 1, w_c, w_s, cnt, False, n)
                if abs(e[ll - 1]) <= thresh:
                    e[ll - 1] = 0.0
        else:
            if idir == 1:
                f = (abs(d[ll - 1]) - shift) * (_sign(1.0, d[ll - 1]) + shift / d[ll - 1])
                g = e[ll - 1]
                for i in range(ll, m):
                    cosr, sinr, r = dlartg(f, g)
                    if i > ll:
                        e[i - 2] = r
                    f = cosr * d[i - 1] + sinr * e[i - 1]
                    e[i - 1] = cosr * e[i - 1] - sinr * d[i - 1]
                    g = sinr * d[i]
                    d[i] = cosr * d[i]
                    cosl, sinl, r = dlartg(f, g)
                    d[i - 1] = r
                    f = cosl * e[i - 1] + sinl * d[i]
                    d[i] = cosl * d[i] - sinl * e[i - 1]
                    if i < m - 1:
                        g = sinl * e[i]
                        e[i] = cosl * e[i]
                    w_c[i - ll], w_s[i - ll], w_oc[i - ll], w_os[i - ll] = cosr, sinr, cosl, sinl
                e[m - 2] = f
                _dlasr_left(VT, ll - 1, w_c, w_s, cnt, True, n)
                _dlasr_right(U, ll - 1, w_oc, w_os, cnt, True, n)
                if abs(e[m - 2]) <= thresh:
                    e[m - 2] = 0.0
            else:
                f = (abs(d[m - 1]) - shift) * (_sign(1.0, d[m - 1]) + shift / d[m - 1])
                g = e[m - 2]
                for i in range(m, ll, -1):
                    cosr, sinr, r = dlartg(f, g)
                    if i < m:
                        e[i - 1] = r
                    f = cosr * d[i - 1] + sinr * e[i - 2]
                    e[i - 2] = cosr * e[i - 2] - sinr * d[i - 1]
                    g = sinr * d[i - 2]
                    d[i - 2] = cosr * d[i - 2]
                    cosl, sinl, r = dlartg(f, g)
                    d[i - 1] = r
                    f = cosl * e[i - 2] + sinl * d[i - 2]
                    d[i - 2] = cosl * d[i - 2] - sinl * e[i - 2]
                    if i > ll + 1:
                        g = sinl * e[i - 3]
                        e[i - 3] = cosl * e[i - 3]
                    w_c[i - ll - 1], w_s[i - ll - 1], w_oc[i - ll - 1], w_os[i - ll - 1] = cosr, -sinr, cosl, -sinl
                e[ll - 1] = f
                if abs(e[ll - 1]) <= thresh:
                    e[ll - 1] = 0.0
                _dlasr_left(VT, ll - 1, w_oc, w_os, cnt, False, n)
                _dlasr_right(U, ll - 1, w_c, w_s, cnt, False, n)
    # make the singular values positive
    for i in range(n):
        if d[i] == 0.0:
            d[i] = 0.0                     # "avoid -ZERO": no flip for a negative zero
        if d[i] < 0.0:
            d[i] = -d[i]
            for k in range(n):
                VT[i][k] = -VT[i][k]
    # sort into decreasing order
    for i in range(1, n):
        isub, smin = 1, d[0]
        for j in range(2, n + 2 - i):
            if d[j - 1] <= smin:
                isub, smin = j, d[j - 1]
        last = n + 1 - i
        if isub != last:
            d[isub - 1] = d[last - 1]
            d[last - 1] = smin
            VT[isub - 1], VT[last - 1] = VT[last - 1], VT[isub - 1]
            for k in range(n):
                U[k][isub - 1], U[k][last - 1] = U[k][last - 1], U[k][isub - 1]


def svd3(A):
    """A: 3 x 3 nested sequence (row-major) -> (U, S, Vh) as nested lists, the way np.linalg.svd returns them."""
    n = 3
    a = [[float(A[r][c]) for c in range(n)] for r in range(n)]
    d, e, tauq, taup = [0.0] * n, [0.0] * (n - 1), [0.0] * n, [0.0] * n
    vq = [None] * n          # Householder vectors of Q below the diagonal (implicit leading 1)
    vp = [None] * n
    # ---- DGEBD2 ----
    for i in range(n):
        beta, tauq[i], v = dlarfg(a[i][i], [a[r][i] for r in range(i + 1, n)])
        d[i] = beta
        vq[i] = [1.0] + v
        for r in range(i + 1, n):
            a[r][i] = v[r - i - 1]
        if i < n - 1 and tauq[i] != 0.0:           # DLARF('Left'): C = A(i:, i+1:)
            for c in range(i + 1, n):
                w = sum(vq[i][r - i] * (a[r][c] if r > i else a[i][c]) for r in range(i, n))
                for r in range(i, n):
                    a[r][c] -= tauq[i] * vq[i][r - i] * w
        if i < n - 1:
            beta, taup[i], v = dlarfg(a[i][i + 1], [a[i][c] for c in range(i + 2, n)])
            e[i] = beta
            vp[i] = [1.0] + v
            for c in range(i + 2, n):
                a[i][c] = v[c - i - 2]
            if taup[i] != 0.0:                      # DLARF('Right'): C = A(i+1:, i+1:)
                for r in range(i + 1, n):
                    w = sum(a[r][c] * vp[i][c - i - 1] for c in range(i + 1, n))
                    for c in range(i + 1, n):
                        a[r][c] -= taup[i] * w * vp[i][c - i - 1]
        else:
            taup[i] = 0.0
    # ---- DBDSDC('U', 'I') -> DLASDQ -> DBDSQR ----
    U = [[1.0 if r == c else 0.0 for c in range(n)] for r in range(n)]
    VT = [[1.0 if r == c else 0.0 for c in range(n)] for r in range(n)]
    dbdsqr(d, e, VT, U, n)
    # ---- DORMBR('Q', 'L', 'N'): U := H(1) H(2) H(3) U  (H(3) first) ----
    for i in range(n - 1, -1, -1):
        if tauq[i] == 0.0:
            continue
        for c in range(n):
            w = sum(vq[i][r - i] * U[r][c] for r in range(i, n))
            for r in range(i, n):
                U[r][c] -= tauq[i] * vq[i][r - i] * w
    # ---- DORMBR('P', 'R', 'T'): VT(:, 2:3) := VT(:, 2:3) G(2) G(1)  (DORML2, right, no transpose: i = k .. 1) ----
    for i in range(n - 2, -1, -1):
        if taup[i] == 0.0:
            continue
        for r in range(n):
            w = sum(VT[r][c] * vp[i][c - i - 1] for c in range(i + 1, n))
            for c in range(i + 1, n):
                VT[r][c] -= taup[i] * w * vp[i][c - i - 1]
    return U, d, VT
