"""TEST INFRASTRUCTURE - NumPy restatement of the reference's SDF voxelisation op.

Follows reference sdf/sdf/csrc/sdf_cuda_kernel.cu line by line (float32 arithmetic, the
double-typed literals of the CUDA source promoted where C would promote them):
  point_segment_distance :73-92      intersect_triangle (Moeller-Trumbore, eps 1e-6) :95-138
  point_triangle_distance :155-237   sdf_cuda_kernel :242-304   launcher :307-335
Voxel (b, k, j, i) (i fastest = x): centre = -1 + (idx + 0.5) * 2/(G-1); phi = min over faces of the
point-triangle distance if the segment towards (-1,-1,-1) crosses an odd number of faces (t >= 0,
unbounded above), else 0.  ``num_faces = faces.shape[0]`` exactly like the launcher (:314): the
reference's caller passes faces as [1, F, 3] (code/utils/fitting.py:367-368), i.e. ONE triangle.

PARITY PINNED against the reference's own kernel: oracle/Makefile compiles sdf_cuda_kernel.cu - unmodified, from
where it lies under /root/reference - for the host (oracle/_ref/libsdf_ref.so; the CUDA qualifiers are defined away by
oracle/ref_shims/, the launcher's thread geometry is replayed by oracle/sdf_ref_driver.cpp), oracle/make_golden_sdf.py
wrote tests/golden/sdf_ref_*.npz with it, and tests/test_sdf_ref.py requires this restatement to reproduce those
fields BIT-EXACTLY (it does: same float32 expression tree, no FMA contraction on either side).  Only the torch / pybind
wrapper of the op (sdf_cuda.cpp, removed ATen APIs) remains unbuildable.
Never imported by the shipped package.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
EPSILON = 0.000001


def _dot(a, b):
    l = np.zeros(a.shape[:-1], F32)
    for i in range(3):
        l = (l + a[..., i] * b[..., i]).astype(F32)
    return l


def _dist(x, y):
    l = np.zeros(np.broadcast_shapes(x.shape, y.shape)[:-1], F32)
    for i in range(3):
        d = (x[..., i] - y[..., i]).astype(F32)
        l = (l + d * d).astype(F32)
    return np.sqrt(l).astype(F32)


def _point_segment(x0, x1, x2):
    dx = (x2 - x1).astype(F32)
    m2 = _dot(dx, dx)
    s12 = ((_dot(x2, dx) - _dot(x0, dx)).astype(F32) / m2).astype(F32)
    s12 = np.where(s12 < 0, F32(0), np.where(s12 > 1, F32(1), s12)).astype(F32)
    r = (s12[..., None] * x1 + (F32(1) - s12)[..., None] * x2).astype(F32)
    return _dist(x0, r), r


def point_triangle_distance(x0, x1, x2, x3):
    """x0 [N,3]; x1..x3 [3].  Returns (distance [N], closest point [N,3])."""
    x13 = (x1 - x3).astype(F32); x23 = (x2 - x3).astype(F32); x03 = (x0 - x3).astype(F32)
    m13 = _dot(x13, x13); m23 = _dot(x23, x23)
    d = _dot(x13, x23)
    invdet = (F32(1.0) / np.maximum((m13 * m23 - d * d).astype(F32), F32(1e-30))).astype(F32)
    a = _dot(np.broadcast_to(x13, x03.shape), x03)
    b = _dot(np.broadcast_to(x23, x03.shape), x03)
    w23 = (invdet * (m23 * a - d * b).astype(F32)).astype(F32)
    w31 = (invdet * (m13 * b - d * a).astype(F32)).astype(F32)
    w12 = (F32(1) - w23 - w31).astype(F32)
    inside = (w23 >= 0) & (w31 >= 0) & (w12 >= 0)
    r_in = (w23[:, None] * x1 + w31[:, None] * x2 + w12[:, None] * x3).astype(F32)
    d_in = _dist(x0, r_in)
    d12, r12 = _point_segment(x0, x1, x2)
    d13, r13 = _point_segment(x0, x1, x3)
    d23, r23 = _point_segment(x0, x2, x3)

    def pick(da, ra, db, rb):
        first = da < db
        return np.where(first, da, db), np.where(first[:, None], ra, rb)
    dA, rA = pick(d12, r12, d13, r13)          # w23 > 0
    dB, rB = pick(d12, r12, d23, r23)          # w31 > 0
    dC, rC = pick(d13, r13, d23, r23)          # else
    c1 = w23 > 0
    c2 = (~c1) & (w31 > 0)
    d_out = np.where(c1, dA, np.where(c2, dB, dC))
    r_out = np.where(c1[:, None], rA, np.where(c2[:, None], rB, rC))
    return (np.where(inside, d_in, d_out).astype(F32),
            np.where(inside[:, None], r_in, r_out).astype(F32))


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F32)


def _dot_macro(a, b):      # the DOT macro: one expression, float32
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]).astype(F32)


def ray_hits(orig, dest, v0, v1, v2):
    """intersect && t >= 0 for the segment direction dest - orig (unnormalised)."""
    dirv = (dest - orig).astype(F32)
    e1 = (v1 - v0).astype(F32); e2 = (v2 - v0).astype(F32)
    pvec = _cross(dirv, np.broadcast_to(e2, dirv.shape))
    det = _dot_macro(np.broadcast_to(e1, pvec.shape), pvec)
    ok = ~((det > -EPSILON) & (det < EPSILON))
    with np.errstate(divide='ignore', invalid='ignore'):
        inv_det = (1.0 / det.astype(np.float64)).astype(F32)            # "1.0 / det": double division, float result
    tvec = (orig - v0).astype(F32)
    # det == 0 (a ray in the triangle's plane, a degenerate triangle) makes inv_det infinite and u / v / t inf or NaN; those
    # voxels are already out through the det test above (the kernel returns there, sdf_cuda_kernel.cu:113-114) - every
    # comparison below is evaluated for them as well only because this is array code: no warnings for arithmetic on them
    with np.errstate(invalid='ignore', over='ignore'):
        u = (_dot_macro(tvec, pvec) * inv_det).astype(F32)
        ok &= ~((u < 0.0) | (u > 1.0))
        qvec = _cross(tvec, np.broadcast_to(e1, tvec.shape))
        v = (_dot_macro(dirv, qvec) * inv_det).astype(F32)
        ok &= ~((v < 0.0) | ((u + v).astype(F32) > 1.0))
        t = (_dot_macro(np.broadcast_to(e2, qvec.shape), qvec) * inv_det).astype(F32)
        return ok & (t >= 0)


def voxel_centres(G):
    dx = F32(2.0 / (G - 1))
    c = (-1 + (np.arange(G) + 0.5) * np.float64(dx)).astype(F32)
    kk, jj, ii = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing='ij')
    return np.stack([c[ii], c[jj], c[kk]], -1).reshape(-1, 3).astype(F32)     # tid order: i fastest


def sdf(faces, vertices, grid_size):
    """faces int [num_faces, 3]; vertices float32 [B, Nv, 3] -> phi [B, G, G, G]."""
    faces = np.asarray(faces).reshape(-1, 3) if np.ndim(faces) == 2 else np.asarray(faces)
    vertices = np.asarray(vertices, F32)
    B, G = vertices.shape[0], grid_size
    cen = voxel_centres(G)
    origin = np.full(3, -1.0, F32)
    phi = np.zeros((B, G * G * G), F32)
    nf = faces.shape[0]
    for bn in range(B):
        mind = np.full(cen.shape[0], 1000.0, F32)
        cnt = np.zeros(cen.shape[0], np.int32)
        for f in range(nf):
            tri = np.asarray(faces[f]).reshape(-1)[:3]
            v1, v2, v3 = (vertices[bn, int(t)] for t in tri)
            _, cp = point_triangle_distance(cen, v1, v2, v3)
            dist = _dist(cen, cp)
            mind = np.where(dist < mind, dist, mind)
            cnt += ray_hits(cen, origin, v1, v2, v3)
        phi[bn] = np.where(cnt % 2 == 0, F32(0), mind)
    return phi.reshape(B, G, G, G)
