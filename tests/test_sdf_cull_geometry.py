"""CPU: the two geometric claims the exact culling of the all-faces SDF term rests on (sdf_term.hip, restated in
oracle/sdf_cull_np.py), checked against the oracle's own per-voxel functions (oracle/sdf_np.py - bit-exact against the
reference's kernel source):
  parity   whenever the float32 ray test (intersect_triangle + t >= 0, sdf_cuda_kernel.cu:95-150) calls a face a hit for a
           corner c, the face is listed in c's projective bin and its nearest vertex is not beyond c - so the count over the
           bin is the count over all faces.  Rays are AIMED at the faces' edges and vertices (inside, on, and just outside
           them, down to 1e-7), incl. slivers of aspect ratio 1000 and grazing incidence: that is where a floating-point
           near miss is called a hit.
  distance a face at distance < r from c has its closest point in a cell that the box c +- r overlaps, and is listed there.
The device kernels are compared with the walk over every face bit for bit in tests/test_gpu_sdf_cull.py / test_gpu_sdf.py;
this file makes the argument checkable without a GPU."""
import numpy as np

from oracle import sdf_cull_np as cu
from oracle import sdf_np

F32 = np.float32
P0 = np.array([-1.0, -1.0, -1.0], F32)


def _triangles(rng, n):
    """n random triangles inside the normalised box |x| <= 0.833: a third regular, a third slivers, a third tiny."""
    ctr = rng.uniform(-0.75, 0.75, (n, 3))
    size = np.where(rng.random(n) < 0.33, rng.uniform(1e-3, 5e-3, n), rng.uniform(0.01, 0.08, n))
    e1 = rng.normal(size=(n, 3)); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = rng.normal(size=(n, 3)); e2 -= (e2 * e1).sum(1, keepdims=True) * e1; e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    aspect = np.where(rng.random(n) < 0.33, 10.0 ** rng.uniform(1, 3, n), 1.0)
    v = np.stack([ctr, ctr + e1 * size[:, None], ctr + (0.3 * e1 + e2 / aspect[:, None]) * size[:, None]], 1)
    return np.clip(v, -0.8333, 0.8333).astype(F32)


def test_a_face_the_ray_test_hits_is_in_the_corners_bin():
    rng = np.random.default_rng(11)
    N = 200000
    tri = _triangles(rng, N)
    # aim: a point of the face's plane at barycentric coordinates scattered around the edges / vertices
    w = rng.dirichlet([0.3, 0.3, 0.3], N)
    w += rng.choice([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3], (N, 3)) * rng.random((N, 3))
    w /= w.sum(1, keepdims=True)
    pt = (w[:, :, None] * tri.astype(np.float64)).sum(1)
    lam = 1.0 + 10.0 ** rng.uniform(-4, 0.5, N)                      # the corner lies behind the face, seen from P
    c = (P0.astype(np.float64) + lam[:, None] * (pt - P0)).astype(F32)
    ok = (c < 1.01).all(1) & (c > -1.0).all(1)
    tri, c = tri[ok], c[ok]
    hit = sdf_np.ray_hits(c, np.broadcast_to(P0, c.shape), tri[:, 0], tri[:, 1], tri[:, 2])
    assert hit.sum() > 20000 and (~hit).sum() > 20000, (hit.sum(), len(hit))     # both outcomes are exercised
    tb = cu.tri_bins(tri)
    assert not tb['bad'].any()
    ia, ib, s_lim = cu.corner_ray_query(c)
    listed = (ia >= tb['a0']) & (ia <= tb['a1']) & (ib >= tb['b0']) & (ib <= tb['b1']) & (tb['min_s'] <= s_lim)
    missed = hit & ~listed
    assert not missed.any(), (int(missed.sum()), np.where(missed)[0][:5])
    # the lists are tight enough to be worth it: a ray that is far from a face does not find it in its bin
    far = rng.permutation(len(c))
    listed_far = (ia[far] >= tb['a0']) & (ia[far] <= tb['a1']) & (ib[far] >= tb['b0']) & (ib[far] <= tb['b1'])
    assert listed_far.mean() < 0.01


def test_a_face_closer_than_r_is_listed_in_a_cell_of_the_query_box():
    rng = np.random.default_rng(12)
    N = 100000
    tri = _triangles(rng, N)
    c = (tri.mean(1) + rng.normal(0, 0.03, (N, 3))).astype(F32)
    # closest point by dense sampling of the face (an upper bound of the distance is all the claim needs)
    u = rng.random((N, 64, 1)); v = rng.random((N, 64, 1)); fl = (u + v) > 1; u = np.where(fl, 1 - u, u); v = np.where(fl, 1 - v, v)
    pts = tri[:, None, 0] * (1 - u - v) + tri[:, None, 1] * u + tri[:, None, 2] * v
    dist = np.linalg.norm(pts - c[:, None], axis=2)
    k = dist.argmin(1)
    near = pts[np.arange(N), k].astype(F32)
    rad = (dist[np.arange(N), k] * 1.001 + 1e-6).astype(F32)           # the radius the kernel would search with
    k0, k1 = cu.corner_cell_query(c, rad)
    tb = cu.tri_bins(tri)
    cell = cu.cell3(near)
    in_query = ((cell >= k0) & (cell <= k1)).all(1)
    in_face = ((cell >= tb['c0']) & (cell <= tb['c1'])).all(1)
    assert in_query.all() and in_face.all(), (int((~in_query).sum()), int((~in_face).sum()))
