"""Writes tests/golden/triangulate.npz from the reference's own recompute3D (code/utils/recompute3D.py) on seeded
synthetic rigs: run in the build container (reference mounted); the GPU box only reads the file."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import                      # noqa: E402
from mvsmplfitting_amd import synthetic as syn     # noqa: E402


def make_case(V, B, seed, drop=0.15, noise=1.5):
    rng = np.random.default_rng(seed)
    cam_R, cam_t, cam_f, cam_c = syn.make_camera_ring(V, radius=3.5 + 0.3 * seed, height=0.3)
    extris = np.tile(np.eye(4), (V, 1, 1))
    extris[:, :3, :3] = cam_R
    extris[:, :3, 3] = cam_t
    intris = np.zeros((V, 3, 3))
    intris[:, 0, 0] = cam_f
    intris[:, 1, 1] = cam_f * (1.0 + 0.01 * rng.normal(size=V))          # fy != fx: the initial guess uses the full K
    intris[:, 0, 2] = cam_c[:, 0]
    intris[:, 1, 2] = cam_c[:, 1]
    intris[:, 2, 2] = 1.0
    X = rng.normal(0, 0.45, (B, 17, 3)) + np.array([0.1, 0.2, 0.0])
    kps = np.zeros((B, V, 17, 3), np.float32)
    for b in range(B):
        for v in range(V):
            p = X[b] @ extris[v, :3, :3].T + extris[v, :3, 3]
            uv = (intris[v] @ p.T).T
            kps[b, v, :, :2] = uv[:, :2] / uv[:, 2:3] + rng.normal(0, noise, (17, 2))
            kps[b, v, :, 2] = rng.uniform(0.2, 1.0, 17)
    kps[..., 2][rng.random((B, V, 17)) < drop] = 0.0                        # undetected joints
    return extris, intris, kps


def main():
    ref_import.load()
    from utils.recompute3D import recompute3D                              # the reference function itself
    out = {}
    for name, (V, B, seed) in dict(v8=(8, 6, 1), v2=(2, 4, 2), v16=(16, 3, 3)).items():
        extris, intris, kps = make_case(V, B, seed)
        ref = np.stack([recompute3D(list(extris), list(intris), [kps[b, v][None].copy() for v in range(V)])
                        for b in range(B)])
        out[name + '_extris'] = extris
        out[name + '_intris'] = intris
        out[name + '_kps'] = kps
        out[name + '_joints3d'] = ref
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'triangulate.npz'), **out)
    print('written', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
