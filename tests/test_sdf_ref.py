"""CPU: pins the SDF voxel function.  tests/golden/sdf_ref_*.npz were written by the REFERENCE's own kernel source
(sdf/sdf/csrc/sdf_cuda_kernel.cu, compiled unmodified for the host: oracle/Makefile -> oracle/_ref/libsdf_ref.so,
generator oracle/make_golden_sdf.py); oracle/sdf_np.py - the NumPy restatement that the on-the-fly term oracle and
older GPU tests use - must reproduce them, and when oracle/_ref is built the goldens are re-derived live."""
import os

import numpy as np
import pytest

from oracle import make_golden_sdf as mg
from oracle import sdf_np, sdf_ref
from tests.helpers import GOLD

CASES = ['wired_g128', 'f64_g32', 'all_g16', 'sphere1_g128', 'sphere_g32', 'sphere_g12']


def load(name):
    g = np.load(os.path.join(GOLD, 'sdf_ref_%s.npz' % name))
    G = int(g['G'])
    shape = (g['verts'].shape[0], G, G, G)
    return g, G, mg.dense(g['idx'], g['val'], shape), mg.dense(g['idx_all'], g['val_all'], shape)


@pytest.mark.skipif(not sdf_ref.available(), reason='oracle/_ref not built (make -C oracle; needs /root/reference)')
@pytest.mark.parametrize('name', CASES)
def test_goldens_are_what_the_reference_kernel_produces(name):
    g, G, phi, phi_all = load(name)
    assert np.array_equal(sdf_ref.sdf(g['faces'], g['verts'], G), phi)
    assert np.array_equal(sdf_ref.sdf(g['faces'], g['verts'], G, all_voxels=True), phi_all)


def test_golden_inputs_are_the_seeded_ones():
    c = mg.cases()
    for name in CASES:
        g = np.load(os.path.join(GOLD, 'sdf_ref_%s.npz' % name))
        assert np.array_equal(g['faces'], c[name]['faces']) and np.array_equal(g['verts'], c[name]['verts']), name


# all_g16 has 13,776 triangles x 8,192 voxels: the NumPy restatement needs ~1 minute for it; the other cases seconds
@pytest.mark.parametrize('name', ['f64_g32', 'sphere1_g128', 'sphere_g32', 'sphere_g12', 'wired_g128'])
def test_numpy_restatement_equals_reference_kernel(name):
    g, G, phi, phi_all = load(name)
    mine = sdf_np.sdf(g['faces'], g['verts'], G)
    assert mine.dtype == np.float32
    # bit-exact: same float32 expression tree, no FMA contraction on either side
    assert np.array_equal(mine, phi_all), (np.abs(mine - phi_all).max(), ((mine > 0) != (phi_all > 0)).sum())


def test_launch_geometry_tail():
    """blocks = B*G^3 / 512 with integer division (sdf_cuda_kernel.cu:317): for G = 12 the last 1728 - 3*512 = 192
    voxels are never written by the reference; they are zero in the stored launch-exact field."""
    g, G, phi, phi_all = load('sphere_g12')
    assert G ** 3 % 512 == 192
    assert not phi.reshape(-1)[3 * 512:].any()
    assert np.array_equal(phi.reshape(-1)[:3 * 512], phi_all.reshape(-1)[:3 * 512])
