"""GPU: the SDF voxelisation op (mvfit_sdf) against the NumPy restatement of the reference's CUDA kernel
(oracle/sdf_np.py, itself bit-exact against the reference kernel source compiled for the host: tests/test_sdf_ref.py).  Inside/outside parity
flips are discontinuous in the inputs, so agreement is asserted as: identical classification
and |delta| <= 1e-5 on >= 99.5 % of the voxels."""
import numpy as np
import pytest
import torch

from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.sdf import SDF
from oracle import sdf_np
from tests.gpu_helpers import make_engine
from tests.helpers import body_model

pytestmark = pytest.mark.gpu


def _mesh(scale, rings=6, segs=8, seed=0):
    v, f = syn._uv_sphere(rings, segs)
    rng = np.random.default_rng(seed)
    v = v * scale * np.array([1.0, 0.8, 0.6]) + rng.normal(0, 0.01, v.shape)      # generic position: no exact edge hits
    return v.astype(np.float32), f.astype(np.int32)


@pytest.mark.parametrize('G', [12, 16])
def test_sdf_matches_restatement(G):
    eng = make_engine(body_model())
    v0, f = _mesh(0.7)
    v1, _ = _mesh(0.5, seed=3)
    verts = np.stack([v0, v1])                                           # B = 2
    ref = sdf_np.sdf(f, verts, G)
    phi = SDF(eng)(torch.tensor(f, device='cuda'), torch.tensor(verts, device='cuda'), grid_size=G).cpu().numpy()
    assert phi.shape == (2, G, G, G)
    same_class = (phi > 0) == (ref > 0)
    close = np.abs(phi - ref) <= 1e-5
    assert (same_class & close).mean() >= 0.995, ((same_class & close).mean(), np.abs(phi - ref).max())
    assert (ref > 0).sum() > 20 and phi.min() >= 0.0                     # fitting.py:369 asserts phi.min() >= 0
    eng.close()


def test_sdf_as_wired_single_triangle_and_errors():
    """The reference's call site hands faces as [1, F, 3] -> num_faces = 1."""
    eng = make_engine(body_model())
    v, f = _mesh(0.7)
    f3 = torch.tensor(f, device='cuda').reshape(1, -1, 3)
    verts = torch.tensor(v[None], device='cuda')
    phi = SDF(eng)(f3, verts, grid_size=16).cpu().numpy()
    ref = sdf_np.sdf(f[:1], v[None], 16)
    assert ((phi > 0) == (ref > 0)).mean() >= 0.999 and np.abs(phi - ref).max() <= 1e-5 + 10 * ((phi > 0) != (ref > 0)).any()
    with pytest.raises(RuntimeError):
        SDF(eng)(f3, verts.cpu(), grid_size=16)                          # CHECK_CUDA
    with pytest.raises(RuntimeError):
        SDF(eng)(f3, verts.expand(2, -1, -1).transpose(1, 2).transpose(1, 2)[:, ::2], grid_size=16)   # CHECK_CONTIGUOUS
    eng.close()


def test_sdf_full_size_properties():
    """G = 128 (the reference's grid), full synthetic body mesh scaled into the unit cube, B = 1: the sign
    structure is scale-consistent and the as-wired single-triangle field is almost empty."""
    model = body_model()
    eng = make_engine(model)
    v = model['v_template'].astype(np.float32)
    c = 0.5 * (v.max(0) + v.min(0))
    s = 1.2 * 0.5 * (v.max(0) - v.min(0)).max()                            # fitting.py:356-363
    vn = ((v - c) / s).astype(np.float32)
    faces = torch.tensor(model['faces'], device='cuda', dtype=torch.int32)
    phi1 = SDF(eng)(faces.reshape(1, -1, 3), torch.tensor(vn[None], device='cuda'), grid_size=128)
    assert phi1.shape == (1, 128, 128, 128) and float(phi1.min()) >= 0.0
    assert float((phi1 > 0).float().mean()) < 0.02                       # one triangle: a thin shadow volume
    phi = SDF(eng)(faces, torch.tensor(vn[None], device='cuda'), grid_size=32)
    inside = float((phi > 0).float().mean())
    assert 0.01 < inside < 0.5 and float(phi.max()) < 1.0
    eng.close()


@pytest.mark.parametrize('name', ['wired_g128', 'f64_g32', 'all_g16', 'sphere1_g128', 'sphere_g32', 'sphere_g12'])
def test_sdf_matches_reference_kernel_goldens(name):
    """mvfit_sdf against fields written by the REFERENCE's own kernel source compiled for the host (oracle/_ref,
    tests/golden/sdf_ref_*.npz, tests/test_sdf_ref.py).  Both sides evaluate the same float32 expression tree without
    FMA contraction, with IEEE division and square root: the comparison is bit-exact, every voxel (the goldens'
    `*_all` field; the reference's launch drops a partial last block, a documented deviation)."""
    import os
    from oracle import make_golden_sdf as mg
    from tests.helpers import GOLD
    g = np.load(os.path.join(GOLD, 'sdf_ref_%s.npz' % name))
    G = int(g['G'])
    ref = mg.dense(g['idx_all'], g['val_all'], (g['verts'].shape[0], G, G, G))
    eng = make_engine(body_model())
    phi = SDF(eng)(torch.tensor(g['faces'], device='cuda'), torch.tensor(g['verts'], device='cuda'), grid_size=G).cpu().numpy()
    eng.close()
    assert phi.shape == ref.shape
    flips = int(((phi > 0) != (ref > 0)).sum())
    assert flips == 0, flips
    assert np.array_equal(phi, ref), (np.abs(phi - ref).max(), int((phi != ref).sum()))


@pytest.mark.parametrize('case', ['body', 'folded', 'edge'])
def test_op_on_face_lists_gives_the_bits_of_the_walk(case):
    """mvfit_sdf with a long face list works on per-call face lists (sdf_term.hip: projective bins for the crossing parity,
    rings of cells for the minimum distance); sdf_face_lists = 0 keeps the walk over every face for every voxel.  Same bits,
    every voxel, G = 128 and all 13,776 faces: a normalised body, a body with folded limbs (self-intersections), and a mesh
    pushed against the -1 faces of the box (the lists do not cover it: its voxels walk all faces)."""
    import os
    import time
    from mvsmplfitting_amd.engine import MvFit
    model = syn.make_body_model(0, skin_topk=4)
    eng = MvFit(model)
    x = np.zeros((1, 118), np.float32)
    x[0, 85] = 1
    if case != 'body':
        x[0, 13:82] = np.random.default_rng(5).normal(0, 0.5, 69)
    eng.set_problems(syn.make_camera_ring(2), np.zeros((1, 2, 17, 2), np.float32), np.ones((1, 2, 17), np.float32))
    v = eng.vertices(x)[0].cpu().numpy()[0]
    c = 0.5 * (v.max(0) + v.min(0))
    s = 1.2 * 0.5 * (v.max(0) - v.min(0)).max()
    vn = ((v - c) / s).astype(np.float32)
    if case == 'edge':
        vn = (vn * 1.15 - 0.03).astype(np.float32)                          # min coordinate ~ -0.99: outside the lists' box
    faces = torch.tensor(model['faces'], device='cuda', dtype=torch.int32)
    vt = torch.tensor(vn[None], device='cuda')
    res, ms = [], []
    for cull in (1, 0):
        eng.set_options(sdf_face_lists=cull)
        SDF(eng)(faces, vt, grid_size=128)
        torch.cuda.synchronize()
        t0 = time.time()
        phi = SDF(eng)(faces, vt, grid_size=128)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.time() - t0))
        res.append(phi.cpu().numpy())
        assert eng.sdf_info()['op'] == ('face_lists' if cull else 'walk'), eng.sdf_info()
    print('mvfit_sdf, 13,776 faces, G = 128, %s: %.2f ms on lists, %.2f ms by the walk' % (case, ms[0], ms[1]))
    assert (res[1] > 0).mean() > 0.005
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
    eng.close()
