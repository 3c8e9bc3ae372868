"""TEST INFRASTRUCTURE - NumPy restatement of the reference's per-view projection of a point set:
``cam(points)`` for every view (reference code/utils/utils.py:581-583, 603-607) with PerspectiveCamera.forward
(code/camera.py:93-117): p = R X + t; uv = f * (p_xy / p_z) + c  (fx == fy = K[0,0], code/init.py:113-119).
Pinned against the reference's own camera class (tests/golden/project.npz, written by oracle/make_golden_project.py;
tests/test_project.py).  Never imported by the shipped package."""
from __future__ import annotations

import numpy as np


def project(points, cams, dtype=np.float64):
    """points [N,3]; cams = (R[V,3,3], t[V,3], f[V], c[V,2]) -> uv [V,N,2]."""
    R, t, f, c = (np.asarray(a, dtype) for a in cams)
    X = np.asarray(points, dtype)
    p = np.einsum('vab,nb->vna', R, X) + t[:, None, :]
    return f[:, None, None] * (p[..., :2] / p[..., 2:3]) + c[:, None, :]
