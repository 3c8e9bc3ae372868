"""Mirror of the reference's per-frame initial guess (code/utils/init_guess.py:18-134 ``init_guess`` for several views,
:190-212 ``fix_params``), batched over the frames of one rig and evaluated by libmvfit:

  recompute3D(extris, intris, keypoints)      code/utils/recompute3D.py:22-62   -> mvfit_triangulate
  umeyama(joints, joints3d, est_scale)        code/utils/umeyama.py:16-109      -> mvfit_umeyama
  cv2.Rodrigues(rot)                          init_guess.py:96                   -> mvfit_umeyama (rvec)
  rest-pose keypoints of the model            init_guess.py:38-52                -> mvfit_vertices at zero parameters
  single-view depth guess                      init_guess.py:54-74                -> mvfit_depth_guess

``recompute3D`` keeps the reference's argument meaning (``keypoints`` a list over views of [1, 17, 3] arrays
(u, v, confidence)); ``init_guess_batch`` takes [B, V, 17, 3]; one view selects the single-view depth guess
(init_guess.py:54-78, ``single_view_joints3d``).

About the rotation: the reference's umeyama evaluates ``U diag(d) Vh^T`` in its full-rank branch (umeyama.py:73), which
depends on the signs LAPACK gave the singular-vector pairs.  mvfit_umeyama's 3 x 3 SVD walks LAPACK's own dgesdd path
(csrc/lapack_svd3.h) and returns np.linalg.svd's pairs, so rotation, translation and scale are the reference's
(tests/test_init_guess_ref.py: the reference's own init_guess on the demo frame, several views and single view)."""
from __future__ import annotations

import numpy as np
import torch

from .engine import MvFit, D

TORSO = (5, 6, 11, 12)          # L / R shoulder, L / R hip in the 17-keypoint order (init_guess.py:57-60,90-92)


def recompute3D_batch(engine: MvFit, extris, intris, keypoints) -> torch.Tensor:
    return engine.triangulate(keypoints, intris, extris)


def recompute3D(engine: MvFit, extris, intris, keypoints) -> np.ndarray:
    assert len(extris) == len(intris) and len(extris) == len(keypoints)          # recompute3D.py:24
    kp = np.concatenate([np.asarray(k, np.float32) for k in keypoints], axis=0)[None]      # [1, V, 17, 3]
    return engine.triangulate(kp, np.asarray(intris), np.asarray(extris))[0].cpu().numpy()


def rest_keypoints(engine: MvFit, scale: float = 1.0) -> torch.Tensor:
    """The 17 keypoints of the model at zero pose / shape / translation and the given scale (init_guess.py:31-52);
    needs set_problems to have been called (any observations)."""
    x = torch.zeros(engine.B, D, device=engine.device)
    x[:, 85] = float(scale)
    _, joints = engine.vertices(x)
    return joints[0].to(torch.float64)


def single_view_joints3d(engine: MvFit, rest, extri, intri, keypoints) -> torch.Tensor:
    """The depth guess for single-view input (init_guess.py:54-74), batched over frames, behind the C ABI
    (mvfit_depth_guess): the rest-pose keypoints pushed along the camera's z axis by est_d = fx * (torso height in 3-D)
    / (torso height in the image), where the 3-D height is the mean of the two shoulder-hip distances and the 2-D one -
    as the reference computes it (:65) - the LEFT shoulder-hip distance taken twice, over the (u, v, confidence) rows.
    rest [17,3] float64, extri [4,4], intri [3,3], keypoints [B,17,3] -> [B,17,3] float64 on the engine's device."""
    return engine.depth_guess(rest, extri, intri, keypoints)


def init_guess_batch(engine: MvFit, extris, intris, keypoints, est_scale=True, fixed_scale=None, use_torso=True,
                     joints3d=None, view_mask=None) -> dict:
    """init_guess (init_guess.py:18-106) for B frames: keypoints [B, V, 17, 3] (u, v, confidence), extris [V,4,4],
    intris [V,3,3] float64.  ``joints3d`` [B,17,3] replaces the triangulation (use_3d, :84-85).
    ``view_mask`` [B, V] bool: the views that exist for a frame.  The reference drops the views without annotation BEFORE
    the initial guess (main.py:44-66), so a frame is triangulated from its own views only (recompute3D would otherwise
    weigh an absent view with its 1e-6, recompute3D.py:47-51) and a frame with exactly one view takes the single-view
    depth guess with that view's camera (:54-78) - whatever the rig's camera count.  Frames are grouped by their view
    pattern, one device call per pattern.
    Returns dict(global_orient [B,3], transl [B,3], scale [B], joints3d [B,17,3]) float64 tensors on the device."""
    kp = np.asarray(keypoints, np.float32) if not isinstance(keypoints, torch.Tensor) else keypoints
    extris, intris = np.asarray(extris, np.float64), np.asarray(intris, np.float64)
    B, V = int(kp.shape[0]), int(kp.shape[1])
    s0 = 1.0 if fixed_scale is None else float(fixed_scale)                      # :24
    rest = rest_keypoints(engine, s0)
    if joints3d is not None:
        j3 = torch.as_tensor(np.asarray(joints3d, np.float64), dtype=torch.float64, device=engine.device)
    else:
        mask = np.ones((B, V), bool) if view_mask is None else np.asarray(view_mask, bool).reshape(B, V)
        j3 = torch.empty(B, 17, 3, dtype=torch.float64, device=engine.device)
        kp_t = torch.as_tensor(kp, dtype=torch.float32)
        for pat in np.unique(mask, axis=0):
            sel = np.flatnonzero((mask == pat[None]).all(1))
            vs = np.flatnonzero(pat)
            if vs.size == 0:
                raise ValueError('frames %s have no view with keypoints' % sel.tolist())
            idx = torch.as_tensor(sel, device=engine.device)
            sub = kp_t[torch.as_tensor(sel)][:, torch.as_tensor(vs)]
            if vs.size == 1:
                j3[idx] = single_view_joints3d(engine, rest, extris[vs[0]], intris[vs[0]], sub[:, 0].contiguous())
            else:
                j3[idx] = engine.triangulate(sub.contiguous(), intris[vs], extris[vs])
    idx = list(TORSO) if use_torso else list(range(17))
    out = engine.umeyama(rest[idx], j3[:, idx].contiguous(), estimate_scale=est_scale)
    scale = out['scale'] if est_scale else torch.full_like(out['scale'], s0)     # :98-101
    return dict(global_orient=out['rvec'], transl=out['trans'], scale=scale, joints3d=j3, rot=out['rot'])


def initial_params(guess: dict, use_vposer: bool, fixed_shape=None) -> torch.Tensor:
    """The flat [B, 118] start of the fit after init_guess + fix_params (init_guess.py:190-212): betas zero (or the
    fixed shape), global_orient / transl / scale from the guess, body_pose = [1, 1, 1, 1, 1, 1, 0 ...] without VPoser
    (:199-203), zero embedding with it (:110-112)."""
    go = guess['global_orient']
    B = go.shape[0]
    x = torch.zeros(B, D, dtype=torch.float32, device=go.device)
    if fixed_shape is not None:
        x[:, 0:10] = torch.as_tensor(np.asarray(fixed_shape, np.float32), device=go.device).reshape(1, 10)
    x[:, 10:13] = go.to(torch.float32)
    if not use_vposer:
        x[:, 13:19] = 1.0
    x[:, 82:85] = guess['transl'].to(torch.float32)
    x[:, 85] = guess['scale'].to(torch.float32)
    return x


__all__ = ['recompute3D', 'recompute3D_batch', 'rest_keypoints', 'single_view_joints3d', 'init_guess_batch', 'initial_params',
           'TORSO']
