"""Sequence mode (reference ``is_seq``) for a batch of sequences.

In the reference a later frame of a sequence starts from the previous frame's result (code/main.py:76-79 ->
``load_init``, code/utils/init_guess.py:137-166), ``non_linear_solver`` then skips the first two stages and scales the
third stage's pose-prior weight by 0.15 (code/utils/non_linear_solver.py:158-162); a frame whose predecessor ended with
a loss above 5000 falls back to the full initial guess and all four stages (init_guess.py:141-145).  Frames of ONE
sequence are therefore a dependent chain; different sequences are independent.  The schedule here is the wavefront
that follows: time step t fits frame t of all S sequences as one batch on the device (S problems per mvfit_fit), the
chain runs along t.  Sequences whose previous loss tripped the 5000 rule are fitted in a second batch with the full
schedule at that step.  (Driving the reference's own caller in sequence mode under patch_reference needs none of this -
the caller's logic does it frame by frame, tests/test_real_caller.py.)"""
from __future__ import annotations

import numpy as np
import torch

from .engine import MvFit, D

RESTART_LOSS = 5000.0          # init_guess.py:142
SEQ_POSE_FACTOR = 0.15         # non_linear_solver.py:162


def sequence_stages(stages):
    """The stage list of a warm-started frame: stages 0 and 1 skipped, stage 2 with body_pose_weight * 0.15 - and the
    bending weight that non_linear_solver derives from it afterwards (:177-179).  The reference holds the weights as
    float32 tensors (non_linear_solver.py:118-124 with float_dtype float32), so `*= 0.15` and `3.17 *` round to float32
    after every operation; the same here, so that the device sees the reference's weights bit for bit
    (tests/test_sequence_ref.py against the weights recorded from the reference)."""
    out = [dict(s) for s in stages[2:]]
    if out:
        w = np.float32(out[0]['body_pose_weight']) * np.float32(SEQ_POSE_FACTOR)
        out[0]['body_pose_weight'] = float(w)
        out[0]['bending_prior_weight'] = float(np.float32(3.17) * w)
    return out


def carry_over(prev_x, prev_loss, x_init_t, use_vposer):
    """The start vector of frame t of every sequence and whether it is a cold start, from frame t - 1's result - the
    reference's load_init + fix_params (code/utils/init_guess.py:137-166,190-215; main.py:76-82):
      * previous loss > 5000 (or no loss returned): init_guess again -> the frame's own full initial guess, all four
        stages (`seq_start = True`, :141-145);
      * otherwise betas, global_orient, transl, scale and - with VPoser - the embedding are the previous frame's result
        (:147-166); the body pose is NOT carried over: reset_params zeroes what it is not given (body_models_scale.py:311-316)
        and fix_params then writes its own start value (:199-203, the six leading ones), which x_init_t holds.
    prev_x [S,118], prev_loss [S], x_init_t [S,118] (NumPy arrays or tensors of one kind) -> (x0 [S,118], cold [S] bool)."""
    is_t = isinstance(prev_x, torch.Tensor)
    pl = prev_loss.detach().cpu().numpy() if is_t else np.asarray(prev_loss)
    cold = ~(pl <= RESTART_LOSS)                     # NaN (no loss returned) restarts too
    x0 = prev_x.clone() if is_t else np.array(prev_x, copy=True)
    if not use_vposer:
        x0[:, 13:82] = x_init_t[:, 13:82]
    if cold.any():
        idx = torch.as_tensor(np.flatnonzero(cold), device=prev_x.device) if is_t else np.flatnonzero(cold)
        x0[idx] = x_init_t[idx]
    return x0, cold


def fit_sequences(engine: MvFit, cams, gt_xy, w_conf, x_init, stages, joints3d=None, **fit_kw):
    """gt_xy [S, T, V, 17, 2], w_conf [S, T, V, 17]: S sequences of T frames of one rig (or per-sequence cameras
    [S, V, ...]); x_init [S, T, 118]: the full initial guess of every frame (used for frame 0 and after a restart).
    joints3d = (gt3d [S, T, 17, 3], conf3d [S, T, 17]): the use_3d targets of every frame (stages carrying F_USE_3D).
    Returns (x [S, T, 118] tensor, dict(final_loss [S, T], n_closure [S, T], restarted [S, T] bool))."""
    gt = np.asarray(gt_xy, np.float32)
    wc = np.asarray(w_conf, np.float32)
    S, T = gt.shape[0], gt.shape[1]
    xi = torch.as_tensor(np.asarray(x_init, np.float32) if not isinstance(x_init, torch.Tensor) else x_init,
                         dtype=torch.float32, device=engine.device).reshape(S, T, D)
    warm = sequence_stages(stages)
    xs = torch.empty(S, T, D, device=engine.device)
    final = torch.empty(S, T, device=engine.device)
    ncl = torch.zeros(S, T, dtype=torch.int32, device=engine.device)
    restarted = np.zeros((S, T), bool)
    use_vposer = bool(int(stages[0].get('flags', 0)) & 1)
    prev_x, prev_loss = None, None
    for t in range(T):
        if t == 0:
            x_start, cold = xi[:, t], np.ones(S, bool)
        else:
            x_start, cold = carry_over(prev_x, prev_loss, xi[:, t], use_vposer)
        restarted[:, t] = cold
        for sel, stg in ((np.flatnonzero(cold), stages), (np.flatnonzero(~cold), warm)):
            if sel.size == 0:
                continue
            cam_sel = tuple(np.asarray(c)[sel] for c in cams) if np.ndim(cams[0]) == 4 else cams
            engine.set_problems(cam_sel, gt[sel, t], wc[sel, t])
            if joints3d is not None:
                engine.set_joints3d(np.asarray(joints3d[0], np.float32)[sel, t], np.asarray(joints3d[1], np.float32)[sel, t])
            idx = torch.as_tensor(sel, device=engine.device)
            xf, st = engine.fit(x_start[idx], stg, **fit_kw)
            xs[idx, t] = xf
            final[idx, t] = st['final_loss']
            ncl[idx, t] = st['n_closure']
        prev_x, prev_loss = xs[:, t], final[:, t]
    return xs, dict(final_loss=final, n_closure=ncl, restarted=restarted)


__all__ = ['fit_sequences', 'sequence_stages', 'carry_over', 'RESTART_LOSS', 'SEQ_POSE_FACTOR']
