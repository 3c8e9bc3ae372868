"""Mirror of the first stage of the reference's per-frame initial guess (code/utils/init_guess.py:80-83):
``recompute3D(extris, intris, keypoints)`` with the reference's argument meaning (code/utils/recompute3D.py:22-62) -
``keypoints`` a list over views of [1, 17, 3] arrays (u, v, confidence) - evaluated by libmvfit
(include/mvfit.h:mvfit_triangulate).  ``recompute3D_batch`` takes [B, V, 17, 3] for a batch of frames of one rig.

The remaining stages of init_guess (Umeyama alignment, cv2.Rodrigues) are not mirrored: the reference's umeyama
multiplies by ``V.T`` of numpy's ``Vh`` (code/utils/umeyama.py:58,73), which makes its rotation depend on LAPACK's
singular-vector sign convention, i.e. it has no implementation-independent value to be equal to."""
from __future__ import annotations

import numpy as np
import torch

from .engine import MvFit


def recompute3D_batch(engine: MvFit, extris, intris, keypoints) -> torch.Tensor:
    return engine.triangulate(keypoints, intris, extris)


def recompute3D(engine: MvFit, extris, intris, keypoints) -> np.ndarray:
    assert len(extris) == len(intris) and len(extris) == len(keypoints)          # recompute3D.py:24
    kp = np.concatenate([np.asarray(k, np.float32) for k in keypoints], axis=0)[None]      # [1, V, 17, 3]
    return engine.triangulate(kp, np.asarray(intris), np.asarray(extris))[0].cpu().numpy()


__all__ = ['recompute3D', 'recompute3D_batch']
