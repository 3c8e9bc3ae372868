"""TEST INFRASTRUCTURE - writes tests/golden/sdf_ref_*.npz from the reference's own SDF kernel (oracle/_ref, built
by `make -C oracle` from /root/reference/sdf/sdf/csrc/sdf_cuda_kernel.cu).  Run in the build container:

    make -C oracle && python -m oracle.make_golden_sdf

Cases (inputs = the seeded synthetic body of mvsmplfitting_amd.synthetic, normalised the way the loss term does it,
code/utils/fitting.py:356-363):
  wired_g128   the reference's call site: faces.reshape(1,-1,3) -> ONE triangle, grid 128, B = 1
  f64_g32      the first 64 triangles, grid 32, B = 2 (rest pose + a posed body)
  all_g16      all 13,776 triangles, grid 16, B = 2
  sphere1_g128 ONE triangle of a small closed mesh at the reference's grid 128 (the as-wired situation with a triangle
               large enough to cast a non-empty shadow: for the body's first triangle wired_g128 is identically zero)
  sphere_g32   the closed 96-triangle mesh at two scales, grid 32, B = 2
  sphere_g12   a small closed mesh, grid 12: 12^3 = 1728 is NOT a multiple of 512 - the reference's launch
               (blocks = total / 512, :317) leaves the last 192 voxels at the caller's zeros; both the launch-exact
               field and the every-voxel field are stored.
phi is stored sparse (indices + values of the non-zero voxels): the fields are mostly zero."""
from __future__ import annotations

import os

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import sdf_ref

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def normalised_bodies():
    """[2, 6890, 3] float32: rest pose and one posed body, (v - centre) / (1.2 * 0.5 * max extent)."""
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(d['rows'], d['cols'], d['vals']))
    orc = cn.ClosureOracle(model, np.float64)
    fr = syn.make_frames(1, seed0=4242)
    posed = orc.body(dict({k: fr[k][0] for k in fr}, use_vposer=False), want_cache=False)['vertices']
    out = []
    for v in (model['v_template'].astype(np.float64), posed):
        v = v.astype(np.float32)
        c = (0.5 * (v.max(0) + v.min(0))).astype(np.float32)
        s = np.float32(1.2 * 0.5) * (v.max(0) - v.min(0)).max()
        out.append(((v - c) / s).astype(np.float32))
    return model, np.stack(out)


def sphere(scale=0.7, seed=0):
    v, f = syn._uv_sphere(6, 8)
    rng = np.random.default_rng(seed)
    v = v * scale * np.array([1.0, 0.8, 0.6]) + rng.normal(0, 0.01, v.shape)      # generic position: no exact edge hits
    return v.astype(np.float32)[None], f.astype(np.int32)


def sparse(phi):
    flat = phi.reshape(-1)
    idx = np.flatnonzero(flat).astype(np.int32)
    return idx, flat[idx].copy()


def dense(idx, val, shape):
    out = np.zeros(int(np.prod(shape)), val.dtype)
    out[idx] = val
    return out.reshape(shape)


def cases():
    model, bodies = normalised_bodies()
    faces = model['faces'].astype(np.int32)
    sv, sf = sphere()
    return {
        'wired_g128': dict(faces=faces[:1], verts=bodies[:1], G=128),
        'f64_g32': dict(faces=faces[:64], verts=bodies, G=32),
        'all_g16': dict(faces=faces, verts=bodies, G=16),
        'sphere1_g128': dict(faces=sf[0:1], verts=sv, G=128),
        'sphere_g32': dict(faces=sf, verts=np.concatenate([sv, sphere(0.5, 3)[0]]), G=32),
        'sphere_g12': dict(faces=sf, verts=sv, G=12),
    }


def main():
    assert sdf_ref.available(), 'build oracle/_ref first: make -C oracle'
    for name, c in cases().items():
        phi = sdf_ref.sdf(c['faces'], c['verts'], c['G'])
        phi_all = sdf_ref.sdf(c['faces'], c['verts'], c['G'], all_voxels=True)
        idx, val = sparse(phi)
        idx_a, val_a = sparse(phi_all)
        np.savez_compressed(os.path.join(GOLD, 'sdf_ref_%s.npz' % name), faces=c['faces'], verts=c['verts'],
                            G=np.int32(c['G']), idx=idx, val=val, idx_all=idx_a, val_all=val_a)
        print(name, 'G', c['G'], 'faces', c['faces'].shape[0], 'B', c['verts'].shape[0], 'nonzero', idx.size,
              'of', phi.size, '(all-voxel launch: %d)' % idx_a.size, 'max', float(phi.max()))


if __name__ == '__main__':
    main()
