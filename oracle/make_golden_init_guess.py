"""TEST INFRASTRUCTURE - writes tests/golden/init_guess_ref.npz: the reference's OWN per-frame initial guess
(code/utils/init_guess.py:18-114 `init_guess`) on the shipped demo's real cameras / keypoints - several views
(recompute3D + umeyama, :80-106) AND the single-view depth guess (:54-78) - SURVEY 8(f) row 1.

    python -m oracle.make_golden_init_guess          (build container; needs /root/reference)

The function is executed unmodified.  Two things it touches do not exist in this container and are bound for the run:
  * `.cuda()` on a CPU tensor (:38) - `torch.Tensor.cuda` returns the tensor itself;
  * `cv2.Rodrigues(rot)[0]` (:96) - cv2 is absent: the stubbed module gets oracle/umeyama_np.py:rotvec (pinned to
    scipy's rotation-vector conversion, tests/test_umeyama.py).
Everything else - the model forward at the reset parameters, joint_regressor / vertex_joint_selector / joint_mapper, the
depth guess arithmetic, recompute3D, umeyama (with LAPACK's singular-vector signs), reset_params - is the reference's.
Body: the seeded synthetic SMPL-shaped body (no SMPL file ships) with the real LSP regressor, float64 like the golden
demo fit.  Stored per case: the inputs and the model's transl / global_orient / scale after the call."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mvsmplfitting_amd import synthetic as syn          # noqa: E402
from oracle import ref_import as ri                      # noqa: E402
from oracle import umeyama_np as un                      # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def main():
    import torch
    ref = ri.load()
    from utils import init_guess as ig
    import cv2                                              # the stub module of oracle/ref_import.py
    cv2.Rodrigues = lambda R: (un.rotvec(np.asarray(R, np.float64)).reshape(3, 1), None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    lsp = ri.real_lsp_regressor()
    model = syn.make_body_model(0, kp_regressor=lsp)
    cams = tuple(g[k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    kp6 = g['keypoints'].reshape(6, 17, 3).astype(np.float64)
    rp = ri.RefProblem(model, cams, g['gt_xy'], g['conf'], 'float64', use_vposer=False)
    cases = {
        'views6': dict(views=[0, 1, 2, 3, 4, 5], fix_scale=False, fixed_scale=None),
        'views3': dict(views=[0, 2, 4], fix_scale=False, fixed_scale=None),
        'views6_fixscale': dict(views=[0, 1, 2, 3, 4, 5], fix_scale=True, fixed_scale=1.3),
        'single0': dict(views=[0], fix_scale=False, fixed_scale=None),
        'single3': dict(views=[3], fix_scale=False, fixed_scale=None),
        'single0_fixscale': dict(views=[0], fix_scale=True, fixed_scale=0.9),
    }
    out = {'model_checksum': np.float64(syn.model_checksum(model))}
    for name, c in cases.items():
        v = c['views']
        setting = dict(model=rp.smpl, dtype=torch.float64, batch_size=1, device=torch.device('cpu'), fix_scale=c['fix_scale'],
                       fixed_scale=c['fixed_scale'], extris=g['extris'][v], intris=g['intris'][v], pose_embedding=None)
        data = {'keypoints': [kp6[i][None] for i in v], '3d_joint': None}
        with torch.no_grad():
            ig.init_guess(setting, data, use_torso=True, model_type='smpllsp', use_vposer=False, use_3d=False)
        p = {k: t.detach().numpy().copy() for k, t in rp.smpl.named_parameters()}
        out[name + '/views'] = np.asarray(v, np.int32)
        out[name + '/fixed_scale'] = np.float64(-1.0 if c['fixed_scale'] is None else c['fixed_scale'])
        out[name + '/transl'] = p['transl'].reshape(3)
        out[name + '/global_orient'] = p['global_orient'].reshape(3)
        out[name + '/scale'] = p['scale'].reshape(())
        print(name, 'transl', p['transl'].reshape(3), 'go', p['global_orient'].reshape(3), 'scale', float(p['scale'].reshape(())))
    np.savez_compressed(os.path.join(GOLD, 'init_guess_ref.npz'), **out)


if __name__ == '__main__':
    main()
