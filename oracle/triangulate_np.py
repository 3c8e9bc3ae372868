"""TEST INFRASTRUCTURE - NumPy restatement of the reference's multi-view linear triangulation
(code/utils/recompute3D.py:22-62, helpers :12-20, get_rot_trans code/utils/utils.py:397-408), the first stage of
the per-frame initial guess (code/utils/init_guess.py:80-83).

For every joint: each view contributes the projector onto the plane orthogonal to its viewing ray,
    n = normalise(K^-1 [u, v, 1]),  N = R^T (I - n n^T),   AtA += (N R)(conf + 1e-6),   Atb += (-N t)(conf + 1e-6)
accumulated in float64; AtA is then rounded to float32 (:54) and the 3x3 system solved in float64 (:59,
np.linalg.solve promotes).  Pinned against the reference function in tests/test_oracle_vs_reference.py and through
the golden vectors tests/golden/triangulate.npz.  Never imported by the shipped package.
"""
from __future__ import annotations

import numpy as np


def recompute3d(extris, intris, keypoints, return_system=False):
    """extris [V,4,4], intris [V,3,3] float64; keypoints [V,17,3] (u, v, confidence) -> joints3d [17,3] float64."""
    extris = np.asarray(extris, np.float64)
    intris = np.asarray(intris, np.float64)
    kp = np.asarray(keypoints).astype(np.float64 if np.asarray(keypoints).dtype == np.float64 else np.float32).copy()
    V, J = kp.shape[0], kp.shape[1]
    conf = kp[:, :, 2].copy()
    kp[:, :, 2] = 1.0
    AtA = np.zeros((J, 3, 3))
    Atb = np.zeros((J, 3))
    for v in range(V):
        Kinv = np.linalg.inv(intris[v])
        R, t = extris[v][:3, :3], extris[v][:3, 3]
        for i in range(J):
            n = Kinv @ kp[v, i]
            n = n / np.linalg.norm(n)
            N = R.T @ (np.eye(3) - np.outer(n, n))
            w = conf[v, i] + 1e-6
            AtA[i] += (N @ R) * w
            Atb[i] += (-N @ t) * w
    AtA = AtA.astype(np.float32)
    out = np.zeros((J, 3))
    for i in range(J):
        out[i] = np.linalg.solve(AtA[i], Atb[i])
    if return_system:
        return out, AtA, Atb
    return out
