"""TEST INFRASTRUCTURE - BASELINE configs[0]: the reference's shipped demo (cfg_files/fit_smpl.yaml) on its REAL inputs.

    python -m oracle.make_golden_demo          (build container; needs /root/reference)

What is real: the six calibrated cameras (data/3DOH50K_Parameters.txt), the six AlphaPose keypoint files
(data/keypoints/0000/Camera00..05/00001_keypoints.json), the image height 1536 (data_weight = 500 / H), the yaml's
weights / optimiser settings (use_vposer: true, body_prior_type l2), and the shipped VPoser checkpoint
priors/snapshots/poser_epoch091.pkl (decoder tensors exported once to tests/golden/vposer_poser_epoch091_decoder.npz,
loaded the way utils/prior.py:38-49 does, map_location='cpu').  What is synthetic: the body (no SMPL file ships,
models/smpl/readme.txt) - the seeded SMPL-shaped body of mvsmplfitting_amd.synthetic with the real LSP regressor.

Everything below is computed by the reference's own code: parsers (utils.load_camera_para, data_parser.read_keypoints),
initial guess (recompute3D + umeyama on the torso, init_guess.py:80-106; cv2.Rodrigues is replaced by scipy's
rotation-vector conversion because cv2 is absent here), closure values and the 4-stage fit
(create_fitting_closure / LBFGSLs / run_fitting, float64 and float32) with every closure call's (x, loss) recorded.
Writes tests/golden/demo_fit_smpl.npz."""
from __future__ import annotations

import glob
import os

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import ref_import as ri
from oracle.make_golden import GOLD, STAGE_POSE_W, STAGE_SHAPE_W, run_reference_fit

REF = ri.REF_ROOT
H_IMG = 1536.0          # data/images/0000/Camera0*/00001.jpg are 2048 x 1536


def export_vposer_decoder():
    """The six decoder tensors of the shipped checkpoint (code/utils/prior.py:38-49; VPoser.py:188-195)."""
    import torch
    ri.load()
    fn = sorted(glob.glob(os.path.join(REF, 'priors', 'snapshots', '*.pkl')))[-1]      # expid2model picks the last snapshot
    full = torch.load(fn, weights_only=False, map_location='cpu')
    sd = full.state_dict()
    names = dict(fc1_w='bodyprior_dec_fc1.weight', fc1_b='bodyprior_dec_fc1.bias', fc2_w='bodyprior_dec_fc2.weight',
                 fc2_b='bodyprior_dec_fc2.bias', out_w='bodyprior_dec_out.weight', out_b='bodyprior_dec_out.bias')
    out = {k: sd[n].detach().cpu().numpy().astype(np.float32) for k, n in names.items()}
    assert out['fc1_w'].shape == (512, 32) and out['fc2_w'].shape == (512, 512) and out['out_w'].shape == (138, 512)
    np.savez_compressed(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz'), **out, source=os.path.basename(fn))
    return out


def demo_inputs():
    ref = ri.load()
    import importlib
    dp = importlib.import_module('utils.data_parser')
    extris, intris = ref.utils.load_camera_para(os.path.join(REF, 'data', '3DOH50K_Parameters.txt'))
    kps = []
    for v in range(6):
        fn = os.path.join(REF, 'data', 'keypoints', '0000', 'Camera%02d' % v, '00001_keypoints.json')
        kt = dp.read_keypoints(fn, use_hands=False, use_face=False)
        kps.append(np.stack(kt.keypoints)[:1])                   # [P, 17, 3] -> the first person (main.py flow)
    trans, rot = ref.utils.get_rot_trans(extris, photoscan=False)
    cams = (np.asarray(rot, np.float64), np.asarray(trans, np.float64), intris[:, 0, 0].copy(), intris[:, :2, 2].copy())
    return extris, intris, kps, cams


def reference_init_guess(rp, extris, intris, kps):
    """init_guess(use_torso=True) (init_guess.py:18-106) + fix_params, with the reference's recompute3D / umeyama."""
    import torch
    from scipy.spatial.transform import Rotation
    from utils.recompute3D import recompute3D
    from utils.umeyama import umeyama
    with torch.no_grad():
        out = rp.smpl(return_verts=True, return_full_pose=True, body_pose=torch.zeros(1, 69, dtype=rp.dt),
                      betas=torch.zeros(1, 10, dtype=rp.dt), global_orient=torch.zeros(1, 3, dtype=rp.dt),
                      transl=torch.zeros(1, 3, dtype=rp.dt))
    joints = out.joints[0].numpy().astype(np.float64)            # scale parameter is 1 at construction
    joints3d = recompute3D(list(extris), list(intris), [k.copy() for k in kps])
    torso = [5, 6, 11, 12]
    rot, trans, scale = umeyama(joints[torso], joints3d[torso], True)
    rvec = Rotation.from_matrix(rot).as_rotvec()
    return dict(joints_rest=joints, joints3d=joints3d, rot=rot, global_orient=rvec, transl=trans, scale=float(scale))


def main():
    from oracle import closure_np as cn
    vpw = export_vposer_decoder()
    extris, intris, kps, cams = demo_inputs()
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(d['rows'], d['cols'], d['vals']))
    gt = np.stack([k[0, :, :2] for k in kps])                    # [V,17,2]
    conf = np.stack([k[0, :, 2] for k in kps])                   # [V,17]
    stages = [dict(data_weight=500.0 / H_IMG, body_pose_weight=STAGE_POSE_W[s], shape_weight=STAGE_SHAPE_W[s],
                   bending_prior_weight=3.17 * STAGE_POSE_W[s], rho=100.0) for s in range(4)]
    out = dict(cam_R=cams[0], cam_t=cams[1], cam_f=cams[2], cam_c=cams[3], gt_xy=gt, conf=conf,
               extris=extris, intris=intris, keypoints=np.stack([k[0] for k in kps]),
               model_checksum=np.array(syn.model_checksum(model)),
               stage_w=np.array([[s['data_weight'], s['body_pose_weight'], s['shape_weight'],
                                  s['bending_prior_weight'], s['rho']] for s in stages]))
    lay, D = cn.param_layout(True)
    for dtn in ('float64', 'float32'):
        rp = ri.RefProblem(model, cams, gt, conf, dtn, use_vposer=True, vposer_weights=vpw)
        if dtn == 'float64':
            ig = reference_init_guess(rp, extris, intris, kps)
            x0 = np.zeros(D)
            x0[lay['global_orient'][0]:lay['global_orient'][1]] = ig['global_orient']
            x0[lay['transl'][0]:lay['transl'][1]] = ig['transl']
            x0[lay['scale'][0]] = ig['scale']
            out.update(x0=x0, init_joints3d=ig['joints3d'], init_joints_rest=ig['joints_rest'], init_rot=ig['rot'])
            print('init guess: scale %.4f transl %s rvec %s' % (ig['scale'], ig['transl'], ig['global_orient']))
            # closure goldens: the initial guess and two points off it (random latent: exercises the decoder's
            # operating region with the REAL weights), first and last stage weights
            rng = np.random.default_rng(91)
            xs = [x0]
            for sig in (0.5, 1.5):
                x = x0.copy()
                x[lay['betas'][0]:lay['betas'][1]] = rng.normal(0, 0.5, 10)
                x[lay['global_orient'][0]:lay['global_orient'][1]] += rng.normal(0, 0.1, 3)
                x[lay['transl'][0]:lay['transl'][1]] += rng.normal(0, 0.05, 3)
                x[lay['pose_embedding'][0]:] = rng.normal(0, sig, 32)
                xs.append(x)
            out['cx'] = np.asarray(xs)
        L, G, J, Vt = [], [], [], []
        for si in (0, 3):
            for x in out['cx']:
                loss, grad, verts, joints = rp.eval_closure(x, stages[si])
                L.append(loss); G.append(grad); J.append(joints); Vt.append(verts)
        sfx = '64' if dtn == 'float64' else '32'
        out['closs' + sfx] = np.asarray(L)
        out['cgrad' + sfx] = np.asarray(G)
        out['cjoints' + sfx] = np.asarray(J)
        if dtn == 'float64':
            out['cverts64_as32'] = np.asarray(Vt)[:3].astype(np.float32)
        rp = ri.RefProblem(model, cams, gt, conf, dtn, use_vposer=True, vposer_weights=vpw)
        final, xf, ncl, trace = run_reference_fit(rp, out['x0'], stages)
        out['fit_final' + sfx] = np.array(final)
        out['fit_xf' + sfx] = xf
        out['fit_ncl' + sfx] = np.array(ncl)
        out['fit_trace' + sfx] = trace[:120]
        print(dtn, 'closure losses', np.asarray(L), 'fit closures/stage', ncl, 'final', final)
    # how much the reference's own float32 fit moves when x0 is perturbed by 1e-6 (relative): the yard-stick for
    # "same quality" on this ill-conditioned problem (the optimum reached is chaotic in the last bits)
    rng = np.random.default_rng(0)
    spread, spread_n = [], []
    for i in range(6):
        x0p = out['x0'] * (1 + rng.normal(0, 1e-6, out['x0'].shape))
        rp = ri.RefProblem(model, cams, gt, conf, 'float32', use_vposer=True, vposer_weights=vpw)
        final, xf, ncl, trace = run_reference_fit(rp, x0p, stages)
        spread.append(final); spread_n.append(ncl)
        print('perturbed x0', i, 'final', final, 'closures/stage', ncl)
    out['fit_spread32'] = np.asarray(spread)
    out['fit_spread_ncl32'] = np.asarray(spread_n)
    np.savez_compressed(os.path.join(GOLD, 'demo_fit_smpl.npz'), **out)


if __name__ == '__main__':
    main()
