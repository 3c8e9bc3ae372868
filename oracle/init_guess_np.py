"""TEST INFRASTRUCTURE - NumPy restatement of the single-view branch of the reference's per-frame initial guess
(code/utils/init_guess.py:54-72), line by line, for ONE frame - including its use of the LEFT shoulder-hip pair twice
in the 2-D height (:65) and the confidence column that `keypoints[0][0][[5, 6, 11, 12]]` (:58) carries into that height.
Pinned: tests/test_init_guess_ref.py checks it (with oracle/umeyama_np.py behind it) against the reference's own
init_guess run on the demo's cameras / keypoints (tests/golden/init_guess_ref.npz, oracle/make_golden_init_guess.py).
Never imported by the package."""
from __future__ import annotations

import numpy as np


def single_view_joints3d(joints, extri, intri, keypoints2d):
    """joints [17,3]: rest-pose keypoints of the model; extri [4,4], intri [3,3]; keypoints2d [17,3] = (u, v, confidence) of the one view."""
    joints = np.asarray(joints, np.float64)
    extri = np.asarray(extri, np.float64)
    torso3d = joints[[5, 6, 11, 12]]                                             # :57
    torso2d = np.asarray(keypoints2d)[[5, 6, 11, 12]]                            # :58
    torso3d = np.insert(torso3d, 3, 1, axis=1).T                                 # :59
    torso3d = (np.dot(extri, torso3d).T)[:, :3]                                  # :60
    diff3d = np.array([torso3d[0] - torso3d[2], torso3d[1] - torso3d[3]])       # :62
    mean_height3d = np.mean(np.sqrt(np.sum(diff3d ** 2, axis=1)))                # :63
    diff2d = np.array([torso2d[0] - torso2d[2], torso2d[0] - torso2d[2]])       # :65 (same pair twice; the rows are
                                                                                 # (u, v, confidence): the confidence difference is in the norm too)
    mean_height2d = np.mean(np.sqrt(np.sum(diff2d ** 2, axis=1)))                # :66
    est_d = np.asarray(intri, np.float64)[0][0] * (mean_height3d / mean_height2d)   # :68
    cam_joints = np.dot(extri, np.insert(joints.copy(), 3, 1, axis=1).T)         # :70-71
    cam_joints[2, :] += est_d                                                    # :72
    return (np.dot(np.linalg.inv(extri), cam_joints).T)[:, :3]                   # :73-74
