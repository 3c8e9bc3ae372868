"""Data formats either side of the fitting path, host side only - so that keypoint files and
camera files of a reference data folder can be turned into the tensors `MvFit.set_problems` / `triangulate` take,
and fitted parameters written the way the reference's tools read them.

  load_camera_para   camera text file: 3-number lines are rows of K, 4-number lines rows of [R|t]; every three rows one
                     matrix, [0,0,0,1] appended to the extrinsics  (reference code/utils/utils.py:352-394)
  read_keypoints     OpenPose-style json {"people":[{"pose_keypoints_2d":[51 floats]}]} -> per person [17,3] float32
                     (reference code/utils/data_parser.py:42-90 with use_hands=False, use_face=False, the
                     'smpllsp' / halpe-17 configuration of fit_smpl.yaml)
  read_joints3d      {"people":[{"pose_keypoints_3d":[...]}]} -> per person [-1,4] float32 (data_parser.py:93-109)
  problem_tensors    one rig + per-view keypoints -> (cams, gt_xy, conf) of MvFit.set_problems, fx for both axes
                     like the reference camera (code/init.py:113-119)
  result_dict        what the reference pickles per person (code/utils/utils.py:744-764, 826-857): the decoded
                     VPoser body_pose has the foot / hand joints zeroed ([18:24], [27:33], [57:]) before it is stored
  save_result_pkl    pickle protocol 2, `<folder>/<serial>/<fn>/000.pkl` (utils.py:859-864)
  save_obj           Wavefront obj, 1-based faces (code/utils/FileLoaders.py:154-160)
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np


def load_camera_para(path):
    intr_rows, extr_rows = [], []
    with open(path, 'r') as f:
        for line in f:
            words = line.strip('\n').rstrip().split()
            if len(words) == 3:
                intr_rows.append([float(w) for w in words])
            elif len(words) == 4:
                extr_rows.append([float(w) for w in words])
    intris = [intr_rows[i:i + 3] for i in range(0, len(intr_rows) - len(intr_rows) % 3, 3)]
    extris = [extr_rows[i:i + 3] + [[0., 0., 0., 1.]] for i in range(0, len(extr_rows) - len(extr_rows) % 3, 3)]
    return np.array(extris), np.array(intris)


def read_keypoints(path):
    with open(path) as f:
        data = json.load(f)
    return [np.array(p['pose_keypoints_2d'], dtype=np.float32).reshape([-1, 3])[:17] for p in data['people']]


def read_joints3d(path):
    with open(path) as f:
        data = json.load(f)
    return [np.array(p['pose_keypoints_3d'], dtype=np.float32).reshape([-1, 4]) for p in data['people']]


def problem_tensors(extris, intris, keypoints_per_view):
    """extris [V,4,4], intris [V,3,3]; keypoints_per_view: list over views of [17,3] (one person, one frame).
    Returns cams = (R[V,3,3], t[V,3], f[V], c[V,2]) float32, gt_xy [1,V,17,2], conf [1,V,17]."""
    extris = np.asarray(extris, np.float64)
    intris = np.asarray(intris, np.float64)
    kp = np.stack([np.asarray(k, np.float32).reshape(-1, 3)[:17] for k in keypoints_per_view])
    cams = (extris[:, :3, :3].astype(np.float32), extris[:, :3, 3].astype(np.float32),
            intris[:, 0, 0].astype(np.float32), intris[:, :2, 2].astype(np.float32))
    return cams, kp[None, :, :, :2].copy(), kp[None, :, :, 2].copy()


def result_dict(x118, loss=None, body_pose_decoded=None):
    """x118: one row of the engine's flat parameter layout (include/mvfit.h); body_pose_decoded [69]: the VPoser
    decode of the fitted embedding when VPoser was used.  Like save_results (utils/utils.py:744-766) the saved body_pose /
    pose have the feet and hand joints zeroed, with or without VPoser."""
    x = np.asarray(x118, np.float32).reshape(-1)
    res = dict(betas=x[0:10][None].copy(), global_orient=x[10:13][None].copy(), transl=x[82:85][None].copy(),
               scale=x[85:86][None].copy())
    if loss is not None:
        res['loss'] = float(loss)
    if body_pose_decoded is not None:
        bp = np.asarray(body_pose_decoded, np.float32).reshape(1, 69).copy()
        res['pose_embedding'] = x[86:118][None].copy()
    else:
        bp = x[13:82][None].copy()
        res['pose_embedding'] = None      # the reference stores the key in both branches (non_linear_solver.py:286: None without VPoser)
    bp[:, 18:24] = 0.          # feet and hands are zeroed in both branches (utils/utils.py:750-753 and :761-764)
    bp[:, 27:33] = 0.
    bp[:, 57:] = 0.
    res['body_pose'] = bp
    res['pose'] = np.hstack((res['global_orient'], bp))
    return res


def save_result_pkl(result_folder, serial, fn, result, person_id=0):
    d = os.path.join(result_folder, serial, fn)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, '{:03d}.pkl'.format(person_id))
    with open(path, 'wb') as f:
        pickle.dump(result, f, protocol=2)
    return path


def save_obj(path, vertices, faces):
    v = np.asarray(vertices).reshape(-1, 3)
    fc = np.asarray(faces).reshape(-1, 3) + 1
    with open(path, 'w') as fp:
        for p in v:
            fp.write('v %f %f %f\n' % (p[0], p[1], p[2]))
        for f in fc:
            fp.write('f %d %d %d\n' % (f[0], f[1], f[2]))


__all__ = ['load_camera_para', 'read_keypoints', 'read_joints3d', 'problem_tensors', 'result_dict',
           'save_result_pkl', 'save_obj']
