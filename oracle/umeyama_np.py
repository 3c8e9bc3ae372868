"""TEST INFRASTRUCTURE - NumPy restatement of the similarity alignment of the reference's per-frame initial guess
(code/utils/init_guess.py:95-106 -> code/utils/umeyama.py:16-109, then cv2.Rodrigues) - SURVEY 8(f) row 1.

The reference's ``umeyama`` is scikit-image's with two local changes that this file restates as they are:
  * full-rank branch (:73): ``T = U diag(d) V.T`` where ``V`` is numpy's ``Vh`` - i.e. ``U diag(d) Vh^T`` instead of
    the Umeyama solution ``U diag(d) Vh``.  That product depends on the SIGNS of the singular-vector pairs, which
    the SVD leaves free (replace (u_k, v_k) by (-u_k, -v_k)): the reference's value is "whatever LAPACK's gesdd
    returned".  ``signs`` (three +-1) selects the pair signs relative to numpy's; (1,1,1) is numpy's own, which is
    what the reference computes in this container.
  * a two-candidate fix (:84-109): the second candidate negates the first two columns of the rotation IN PLACE in
    ``T`` (``rot`` is a view), the one with the smaller alignment residual is returned - and the translation is
    computed from ``T`` after the loop, i.e. always with the SECOND candidate's rotation (:104).
Pinned against the reference function itself (tests/test_umeyama.py).  ``rotvec`` restates cv2.Rodrigues
(matrix -> rotation vector); cv2 is absent here, it is pinned to scipy's conversion.  Never imported by the package."""
from __future__ import annotations

import numpy as np

SIGN_PATTERNS = [(1, 1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, -1)]        # modulo a global sign, which cancels


def umeyama(src, dst, estimate_scale=True, signs=(1, 1, 1)):
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    num, dim = src.shape
    src_mean, dst_mean = src.mean(0), dst.mean(0)
    sd, dd = src - src_mean, dst - dst_mean
    A = dd.T @ sd / num                                                    # :45
    d = np.ones(dim)
    if np.linalg.det(A) < 0:                                               # :48-50
        d[dim - 1] = -1
    U, S, Vh = np.linalg.svd(A)
    D = np.diag(np.asarray(signs, np.float64))
    U, Vh = U @ D, D @ Vh                                                  # an equally valid SVD of A
    rank = np.linalg.matrix_rank(A)
    if rank == 0:
        return None
    if rank == dim - 1:                                                    # :60-68 (uses Vh: the textbook formula)
        if np.linalg.det(U) * np.linalg.det(Vh) > 0:
            T = U @ Vh
        else:
            d2 = d.copy()
            d2[dim - 1] = -1
            T = U @ np.diag(d2) @ Vh
    else:
        T = U @ np.diag(d) @ Vh.T                                          # :73
    scale = 1.0 / sd.var(axis=0).sum() * (S @ d) if estimate_scale else 1.0
    rots, losses = [], []
    rot = T.copy()
    for i in range(2):                                                     # :84-98
        if i == 1:
            rot[:, :2] *= -1
        t_i = dst_mean - scale * rot @ src_mean
        transed = (scale * rot @ src.T).T + t_i
        losses.append(np.linalg.norm(transed - dst))
        rots.append(rot.copy())
    trans = dst_mean - scale * rots[1] @ src_mean                          # :104 (T was flipped in place)
    return (rots[1] if losses[0] > losses[1] else rots[0]), trans, scale, losses


def rotvec(R):
    """cv2.Rodrigues(R)[0] for a rotation matrix: theta * axis from the antisymmetric part; theta near pi from the
    diagonal."""
    R = np.asarray(R, np.float64)
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r * r).sum() * 0.25)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        x = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        y = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        z = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (theta / np.linalg.norm(v))
    return r * (0.5 / s) * theta
