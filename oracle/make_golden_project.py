"""TEST INFRASTRUCTURE - tests/golden/project.npz from the reference's own PerspectiveCamera (code/camera.py:93-117):
the per-view projection of full meshes (visualize_fitting, code/utils/utils.py:603-607) on the demo's six real
cameras and on an 8-camera synthetic ring.   python -m oracle.make_golden_project   (build container)"""
from __future__ import annotations

import os

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import ref_import as ri
from oracle.make_golden import GOLD


def ref_project(points, cams, dtype):
    import torch
    ref = ri.load()
    dt = torch.float64 if dtype == 'float64' else torch.float32
    out = []
    for v in range(cams[0].shape[0]):
        cam = ref.camera.create_camera(focal_length_x=float(cams[2][v]), focal_length_y=float(cams[2][v]),
                                       translation=torch.tensor(cams[1][v], dtype=dt).unsqueeze(0),
                                       rotation=torch.tensor(cams[0][v], dtype=dt).unsqueeze(0),
                                       center=torch.tensor(cams[3][v], dtype=dt).unsqueeze(0), dtype=dt)
        with torch.no_grad():
            out.append(cam(torch.tensor(points, dtype=dt).unsqueeze(0))[0].numpy())
    return np.stack(out)


def main():
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(d['rows'], d['cols'], d['vals']))
    demo = np.load(os.path.join(GOLD, 'demo_fit_smpl.npz'))
    rng = np.random.default_rng(5)
    out = {}
    # (1) the demo rig: the body scaled / placed like the demo's initial guess
    cams = tuple(demo[k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    x0 = demo['x0']
    pts = (model['v_template'].astype(np.float64) * x0[16] + x0[13:16] + rng.normal(0, 0.01, (6890, 3))).astype(np.float32)
    out.update(demo_cam_R=cams[0], demo_cam_t=cams[1], demo_cam_f=cams[2], demo_cam_c=cams[3], demo_pts=pts,
               demo_uv64=ref_project(pts.astype(np.float64), cams, 'float64'),
               demo_uv32=ref_project(pts, tuple(a.astype(np.float32) for a in cams), 'float32'))
    # (2) the synthetic 8-camera ring, two bodies
    ring = syn.make_camera_ring(8)
    pts2 = np.stack([model['v_template'] + rng.normal(0, 0.02, (6890, 3)) for _ in range(2)]).astype(np.float32)
    out.update(ring_pts=pts2,
               ring_uv64=np.stack([ref_project(p.astype(np.float64), ring, 'float64') for p in pts2]),
               ring_uv32=np.stack([ref_project(p, ring, 'float32') for p in pts2]))
    # keep the file small: the full point sets, the reference's pixels for every 8th point
    idx = np.arange(0, 6890, 8)
    for k in ('demo_uv64', 'demo_uv32'):
        out[k] = out[k][:, idx]
    for k in ('ring_uv64', 'ring_uv32'):
        out[k] = out[k][:, :, idx]
    out['idx'] = idx
    np.savez_compressed(os.path.join(GOLD, 'project.npz'), **out)
    for k in ('demo_uv64', 'ring_uv64'):
        print(k, out[k].shape, float(np.abs(out[k]).max()),
              'fp32-vs-fp64 max abs', float(np.abs(out[k] - out[k.replace('64', '32')]).max()))


if __name__ == '__main__':
    main()
