"""File-to-file batch driver: what the reference's `main.py` does for a data folder (code/main.py:21-130) with every
frame of a serial fitted in ONE batched device fit (the reference walks the frames one by one):

    keypoint files  <keyp_root>/<serial>/<camera>/<frame>_keypoints.json   (data_parser.py:375-411: cameras and frames
                    in sorted order, view v = the v-th camera of the camera file; a view without a file does not take
                    part in that frame (main.py:44-66) - here: zero confidence in the fit, and left out of that frame's
                    initial guess: triangulation over the frame's own views, the single-view depth guess when it has one)
    camera file     io_formats.load_camera_para  (utils.py:352-394)
      -> joint weights: hips 11, 12 ignored unless pose_format == 'lsp14' and use_hip (data_parser.py:340-357;
         model_type 'smpllsp' -> 'lsp14', init.py:63-69, and fit_smpl.yaml has use_hip: true: the default here)
      -> initial guess on the device (init_guess.py:18-106 + fix_params :190-212): init_guess.init_guess_batch
      -> staged fit of all frames at once (non_linear_solver.py:156-211), or the warm-start chain with is_seq
         (main.py:76-79, init_guess.py:137-166): sequence.fit_sequences
      -> per frame `<result_folder>/<serial>/<frame>/000.pkl` (utils.py:744-766, 859-864: pickle protocol 2, body pose
         decoded and feet / hands zeroed) and, with save_meshes, `<mesh_folder>/<serial>/<frame>/000.obj` of the model at the
         SAVED (zeroed) pose (utils.py:866-890).

Images, rendering and the interactive viewer of main.py are out of scope; nothing here reads images, so
the image height the data weight 500 / H refers to (non_linear_solver.py:150,177) is an argument.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from . import _lib
from . import io_formats as iof
from .engine import MvFit, stage_weights
from .init_guess import init_guess_batch, initial_params
from .sequence import fit_sequences


def list_frames(keyp_root):
    """[(serial, [camera names], [(frame name, [json path or None per camera])])] in the reference's order."""
    out = []
    for serial in sorted(os.listdir(keyp_root)):
        sdir = os.path.join(keyp_root, serial)
        if not os.path.isdir(sdir):
            continue
        cams = sorted(c for c in os.listdir(sdir) if os.path.isdir(os.path.join(sdir, c)))
        names = set()
        for c in cams:
            names.update(f[:-len('_keypoints.json')] for f in os.listdir(os.path.join(sdir, c)) if f.endswith('_keypoints.json'))
        frames = []
        for fn in sorted(names):
            paths = [os.path.join(sdir, c, fn + '_keypoints.json') for c in cams]
            frames.append((fn, [p if os.path.exists(p) else None for p in paths]))
        out.append((serial, cams, frames))
    return out


def load_serial(frames, num_views, person=0, return_mask=False):
    """Keypoints [F, V, 17, 3] float32 of one serial (missing view / person: zeros, i.e. zero confidence) and, with
    return_mask, which (frame, view) pairs exist - the reference drops the others from the frame (main.py:44-66)."""
    kp = np.zeros((len(frames), num_views, 17, 3), np.float32)
    mask = np.zeros((len(frames), num_views), bool)
    for f, (_, paths) in enumerate(frames):
        for v, p in enumerate(paths[:num_views]):
            if p is None:
                continue
            people = iof.read_keypoints(p)
            if len(people) > person:
                kp[f, v] = people[person]
                mask[f, v] = True
    return (kp, mask) if return_mask else kp


def load_serial_joints3d(frames, person=0):
    """use_3d annotation (data_parser.py:396-400): per frame the 'pose_keypoints_3d' of the FIRST camera that has a file,
    [F, 17, 4] = (x, y, z, confidence) float32, and a mask of the frames that carry one."""
    j3 = np.zeros((len(frames), 17, 4), np.float32)
    has = np.zeros(len(frames), bool)
    for f, (_, paths) in enumerate(frames):
        for p in paths:
            if p is None:
                continue
            try:
                people = iof.read_joints3d(p)
            except KeyError:                                   # no 3-D annotation in this file (the reference: None)
                break
            if len(people) > person:
                j3[f] = people[person][:17]
                has[f] = True
            break
    return j3, has


def fit_folder(model: dict, keyp_root, cam_file, result_folder, *, vposer=None, image_height=1536.0, is_seq=False,
               pose_format='lsp14', use_hip=True, use_3d=False, fix_scale=None, fix_shape=None, save_meshes=False,
               mesh_folder=None, device=0, stages=None, engine: MvFit | None = None, timing: dict | None = None):
    """Fits every frame under keyp_root and writes the reference's result files.  Returns
    {serial: dict(frames, params [F,118], final_loss [F], n_closure [F], files [F], init [F,118], restarted [F]:
    frames fitted from their own initial guess - all of them unless is_seq, used_3d [F]: frames fitted with the 3-D joint
    term, views_per_frame [F])}.  ``timing``: a dict that receives the wall-clock seconds of the four steps of the pipeline
    (read = directory walk + keypoint / camera files, init_guess, fit, write = decoded pose + result files [+ meshes]),
    summed over the serials - the end-to-end figure next to the reference's only timer (code/main.py:27,91-94)."""
    import time as _time

    def _tick(key, t0):
        if timing is not None:
            torch.cuda.synchronize(eng.device)
            timing[key] = timing.get(key, 0.0) + (_time.time() - t0)
        return _time.time()
    _t = _time.time()
    extris, intris = iof.load_camera_para(cam_file)
    use_vposer = vposer is not None
    flags = _lib.F_VPOSER if use_vposer else 0
    if fix_scale is not None:
        flags |= _lib.F_FIX_SCALE
    if fix_shape is not None:
        flags |= _lib.F_FIX_SHAPE
    user_stages = stages
    own = engine is None
    eng = engine if engine is not None else MvFit(model, vposer=vposer, device=device)
    jw = np.ones(17, np.float32)
    if pose_format != 'lsp14' or not use_hip:                      # data_parser.py:353-356
        jw[11] = jw[12] = 0.0
    results = {}
    try:
        _t = _tick('read', _t)
        for serial, cams, frames in list_frames(keyp_root):
            V, F = len(cams), len(frames)
            if F == 0 or V < 1:
                continue
            if V > len(extris):
                raise ValueError('serial %s has %d camera folders, the camera file %s holds %d cameras' % (serial, V, cam_file, len(extris)))
            kp, vmask = load_serial(frames, V, return_mask=True)
            if not vmask.any(1).all():
                raise ValueError('serial %s: frames %s have no keypoint file in any camera folder'
                                 % (serial, [frames[f][0] for f in np.flatnonzero(~vmask.any(1))]))
            ex, it = np.asarray(extris[:V], np.float64), np.asarray(intris[:V], np.float64)
            rig = (ex[:, :3, :3].astype(np.float32), ex[:, :3, 3].astype(np.float32),
                   it[:, 0, 0].astype(np.float32), it[:, :2, 2].astype(np.float32))
            gt_xy = kp[..., :2].copy()
            conf = kp[..., 2] * jw[None, None, :]
            eng.set_problems(rig, gt_xy, conf)
            _t = _tick('read', _t)
            # 3-D joint targets (non_linear_solver.py:86-99) and the initial alignment to them instead of the triangulation
            # (init_guess.py:84-85) - decided PER FRAME like the reference (:68-69): frames with an annotation are fitted with
            # the 3-D term, the others without it (two batched fits when a serial mixes both)
            ann, has = (load_serial_joints3d(frames) if use_3d else (None, np.zeros(F, bool)))
            if is_seq and has.any() and not has.all():
                warnings.warn('serial %s: %d of %d frames carry no 3-D annotation; the is_seq chain runs on one objective - '
                              'fitting the whole serial from the 2-D keypoints only' % (serial, int((~has).sum()), F))
                has[:] = False
            c3 = None
            if has.any():
                c3 = ann[:, :, 3].copy()
                if not use_hip:
                    c3[:, 11] = c3[:, 12] = 0.0
            j3_all = ann[:, :, :3].astype(np.float64) if has.all() else None
            guess = init_guess_batch(eng, ex, it, kp, est_scale=fix_scale is None, fixed_scale=fix_scale, joints3d=j3_all,
                                     view_mask=vmask)
            if has.any() and not has.all():
                # annotated frames: aligned to their 3-D joints; the rest keep the triangulation / depth guess
                sel = np.flatnonzero(has)
                eng.set_problems(rig, gt_xy[sel], conf[sel])
                g3 = init_guess_batch(eng, ex, it, kp[sel], est_scale=fix_scale is None, fixed_scale=fix_scale,
                                      joints3d=ann[sel][:, :, :3].astype(np.float64))
                idx = torch.as_tensor(sel, device=eng.device)
                for k in ('global_orient', 'transl', 'scale', 'joints3d', 'rot'):
                    guess[k][idx] = g3[k]
                eng.set_problems(rig, gt_xy, conf)
            x0 = initial_params(guess, use_vposer, fixed_shape=fix_shape)
            _t = _tick('init_guess', _t)

            def stages_for(with_3d):
                if user_stages is None:
                    return stage_weights(float(image_height), flags=flags | (_lib.F_USE_3D if with_3d else 0))
                # caller's stage list: the 3-D term follows the group being fitted (annotated frames carry it, the others
                # must not run it against absent targets), whatever the caller's flag words say
                out = []
                for st_ in user_stages:
                    st_ = dict(st_)
                    f_ = int(st_.get('flags', 0))
                    st_['flags'] = (f_ | _lib.F_USE_3D) if with_3d else (f_ & ~_lib.F_USE_3D)
                    out.append(st_)
                return out
            if is_seq:
                if has.all():
                    eng.set_joints3d(ann[:, :, :3], c3)
                t3 = (ann[None, :, :, :3], c3[None]) if has.all() else None
                xs, st = fit_sequences(eng, rig, gt_xy[None], conf[None], x0[None], stages_for(has.all()), joints3d=t3)
                xf, final, ncl = xs[0], st['final_loss'][0], st['n_closure'][0]
                restarted = st['restarted'][0]
                eng.set_problems(rig, gt_xy, conf)                  # back to the whole serial for the outputs below
            else:
                xf = torch.empty_like(x0)
                final = torch.empty(F, device=eng.device)
                ncl = torch.zeros(F, dtype=torch.int32, device=eng.device)
                for sel, with_3d in ((np.flatnonzero(has), True), (np.flatnonzero(~has), False)):
                    if sel.size == 0:
                        continue
                    if sel.size < F:
                        eng.set_problems(rig, gt_xy[sel], conf[sel])
                    if with_3d:
                        eng.set_joints3d(ann[sel][:, :, :3], c3[sel])
                    idx = torch.as_tensor(sel, device=eng.device)
                    xs_, st = eng.fit(x0[idx], stages_for(with_3d))
                    xf[idx], final[idx], ncl[idx] = xs_.to(xf.dtype), st['final_loss'].to(final.dtype), st['n_closure'].to(ncl.dtype)
                if has.any() and not has.all():
                    eng.set_problems(rig, gt_xy, conf)
                restarted = np.ones(F, bool)
            _t = _tick('fit', _t)
            full = eng.full_pose(xf, flags=flags & ~_lib.F_USE_3D).cpu().numpy()
            xf_h, final_h = xf.cpu().numpy(), final.cpu().numpy()
            res = [iof.result_dict(xf_h[f], loss=final_h[f], body_pose_decoded=full[f, 3:] if use_vposer else None)
                   for f in range(F)]
            files = [iof.save_result_pkl(result_folder, serial, frames[f][0], res[f]) for f in range(F)]
            if save_meshes:
                # the mesh of the SAVED parameters: zeroed feet / hands, model(global_orient, transl, body_pose, betas)
                xm = xf_h.copy()
                xm[:, 13:82] = np.stack([r['body_pose'][0] for r in res])
                verts, _ = eng.vertices(xm, flags=flags & ~_lib.F_VPOSER)
                verts = verts.cpu().numpy()
                for f in range(F):
                    d = os.path.join(mesh_folder or os.path.join(result_folder, 'meshes'), serial, frames[f][0])
                    os.makedirs(d, exist_ok=True)
                    iof.save_obj(os.path.join(d, '000.obj'), verts[f], model['faces'])
            results[serial] = dict(frames=[fr[0] for fr in frames], params=xf_h, final_loss=final_h,
                                   n_closure=ncl.cpu().numpy(), files=files, init=x0.cpu().numpy(), restarted=restarted,
                                   used_3d=has.copy(), views_per_frame=vmask.sum(1))
            _t = _tick('write', _t)
    finally:
        if own:
            eng.close()
    return results


__all__ = ['list_frames', 'load_serial', 'fit_folder']
