"""TEST INFRASTRUCTURE - NumPy restatement of the SDF interpenetration term of SMPLifyLoss.forward
(reference code/utils/fitting.py:282-288 bounding boxes, :352-393 the term) with a hand-derived adjoint.

    boxes = [min_v, max_v] per axis;  c = mean(boxes);  s = (1 + 0.2) * 0.5 * max_axis(max - min)
    phi   = SDF(faces.reshape(1, -1, 3), (v - c) / s, grid 128)             (no_grad; float32 op)
    phi_v = grid_sample(phi, (v - c) / s)      trilinear, zeros padding, align_corners=False
    pen   = (coll_loss_weight * sum_v phi_v / valid_people)^2,   valid_people == 1 (fitting.py:366)

As wired the SDF op receives faces of shape [1, F, 3], so it voxelises the FIRST triangle only
(oracle/sdf_np.py header); ``num_faces`` reproduces that (1) or any other prefix of the face list.
Gradient paths: through the sampling coordinates ((v - c) / s of every vertex, c and s through the arg-min /
arg-max vertices of the bounding box); phi itself is a constant.

Parity pins: the voxel values come from oracle/sdf_np.py, which is bit-exact against the reference's own kernel source
compiled for the host (oracle/_ref, tests/test_sdf_ref.py); the sampling + bounding-box part is pinned against
torch.nn.functional.grid_sample + autograd in tests/test_oracle_sdf_term.py.
Never imported by the shipped package.
"""
from __future__ import annotations

import numpy as np

from oracle import sdf_np

BOX_MARGIN = (1 + 0.2) * 0.5       # a Python double; cast to the tensor's dtype when it meets the tensor


def bounding_box(verts):
    """min / max per axis with first-occurrence arg indices (fitting.py:282-288)."""
    imin = verts.argmin(0)
    imax = verts.argmax(0)
    return verts[imin, np.arange(3)], verts[imax, np.arange(3)], imin, imax


def normalise(verts, dtype):
    v = np.asarray(verts, dtype)
    lo, hi, imin, imax = bounding_box(v)
    c = (lo + hi) / dtype(2)                                     # boxes.mean(dim=1)
    ext = hi - lo
    amax = int(ext.argmax())
    s = dtype(BOX_MARGIN) * ext[amax]
    return c, s, imin, imax, amax


def sample_trilinear(phi, loc):
    """grid_sample(phi[None, None], loc.view(1, -1, 1, 1, 3)): values and d value / d loc.
    phi [G, G, G] indexed [z, y, x]; loc [..., (x, y, z)] in [-1, 1]."""
    G = phi.shape[0]
    pix = ((loc + 1) * G - 1) / 2                                # align_corners=False unnormalisation
    i0 = np.floor(pix).astype(np.int64)
    fr = pix - i0
    val = np.zeros(loc.shape[0], loc.dtype)
    gpix = np.zeros_like(loc)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ix, iy, iz = i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz
                ok = (ix >= 0) & (ix < G) & (iy >= 0) & (iy < G) & (iz >= 0) & (iz < G)
                p = np.where(ok, phi[np.clip(iz, 0, G - 1), np.clip(iy, 0, G - 1), np.clip(ix, 0, G - 1)], 0).astype(loc.dtype)
                wx = fr[:, 0] if dx else 1 - fr[:, 0]
                wy = fr[:, 1] if dy else 1 - fr[:, 1]
                wz = fr[:, 2] if dz else 1 - fr[:, 2]
                val += p * wx * wy * wz
                gpix[:, 0] += p * (1 if dx else -1) * wy * wz
                gpix[:, 1] += p * wx * (1 if dy else -1) * wz
                gpix[:, 2] += p * wx * wy * (1 if dz else -1)
    return val, gpix * (G / 2)


def sdf_term(verts, faces, coll_w, num_faces=1, grid_size=128, dtype=np.float64, phi=None):
    """pen, d pen / d verts [Nv,3], aux.   verts [Nv,3] (model vertices incl. translation)."""
    v = np.asarray(verts, dtype)
    c, s, imin, imax, amax = normalise(v, dtype)
    if phi is None:
        # the op runs in float32 on the float32 normalised vertices (fitting.py:362-368)
        v32 = np.asarray(verts, np.float32)
        c32, s32, *_ = normalise(v32, np.float32)
        vn = ((v32 - c32) / s32).astype(np.float32)
        phi = sdf_np.sdf(np.asarray(faces).reshape(-1, 3)[:num_faces], vn[None], grid_size)[0]
    loc = (v - c) / s
    val, gloc = sample_trilinear(np.asarray(phi, dtype), loc)
    S = val.sum()
    pen = (dtype(coll_w) * S) ** 2
    fac = 2 * dtype(coll_w) ** 2 * S
    # adjoint of S w.r.t. the vertices
    g = gloc / s
    g_c = -g.sum(0)
    g_s = -(gloc * loc).sum() / s
    for a in range(3):
        g[imin[a], a] += g_c[a] / 2
        g[imax[a], a] += g_c[a] / 2
    g[imax[amax], amax] += dtype(BOX_MARGIN) * g_s
    g[imin[amax], amax] -= dtype(BOX_MARGIN) * g_s
    return pen, fac * g, dict(S=S, phi=phi, c=c, s=s, dS_dverts=g, phi_val=val)
