"""TEST INFRASTRUCTURE - NumPy restatement of the face binning of the all-faces SDF term (mvsmplfitting_amd/csrc/sdf_term.hip:
sdf_bin2 / sdf_cell3 / sdf_tri_bins and the queries of sdf_sample_culled_kernel / sdf_voxelize_culled_kernel), float32 like
the device code.  Used by tests/test_sdf_cull_geometry.py to check the two claims the culling rests on against the oracle's
own per-voxel functions (oracle/sdf_np.py, bit-exact against the reference kernel): a face the ray test calls a hit is in the
corner's projective bin and not beyond the corner; a face closer than r has a point in a cell the box corner +- r overlaps."""
import numpy as np

F32 = np.float32
NB = 256
NC3 = 64
DELTA = F32(2e-3)


def bin2(a):
    return np.clip((np.asarray(a, F32) * F32(NB)).astype(np.int64), 0, NB - 1)


def cell3(x):
    return np.clip(((np.asarray(x, F32) + F32(1.0)) * F32(0.5 * NC3)).astype(np.int64), 0, NC3 - 1)


def tri_bins(p):
    """p [N, 3, 3] float32 normalised vertices -> dict of the bin / cell ranges and min_s, `bad` flags."""
    p = np.asarray(p, F32)
    q = p + F32(1.0)
    sq = (q[..., 0] + q[..., 1]) + q[..., 2]
    bad = ~((q > F32(8.0) * DELTA).all(-1) & (sq < F32(16.0))).all(-1)
    amin = ((q[..., 0] - DELTA) / (sq + DELTA)).min(-1)
    amax = ((q[..., 0] + DELTA) / (sq - DELTA)).max(-1)
    bmin = ((q[..., 1] - DELTA) / (sq + DELTA)).min(-1)
    bmax = ((q[..., 1] + DELTA) / (sq - DELTA)).max(-1)
    return dict(a0=bin2(amin), a1=bin2(amax), b0=bin2(bmin), b1=bin2(bmax), min_s=sq.min(-1),
                c0=cell3(p.min(1)), c1=cell3(p.max(1)), bad=bad)


def corner_ray_query(c):
    """c [N, 3] float32 -> (bin_a, bin_b, depth limit) of the parity query."""
    c = np.asarray(c, F32)
    q = c + F32(1.0)
    sq = (q[:, 0] + q[:, 1]) + q[:, 2]
    return bin2(q[:, 0] / sq), bin2(q[:, 1] / sq), sq + F32(3.0) * DELTA


def corner_cell_query(c, rad):
    """cells the box c +- rad overlaps: (k0 [N, 3], k1 [N, 3])."""
    c = np.asarray(c, F32)
    rad = np.asarray(rad, F32)[:, None]
    return cell3(c - rad), cell3(c + rad)
