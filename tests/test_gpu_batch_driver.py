"""GPU: the file-to-file batch driver (mvsmplfitting_amd/batch.py) on the reference's shipped demo
inputs (tests/golden/demo_data): keypoint / camera files -> initial guess on the device -> staged fit -> the reference's
result files."""
import os
import pickle
import shutil

import numpy as np
import pytest

from mvsmplfitting_amd import batch
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLD, 'demo_data')
G = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))


def _vposer():
    d = dict(np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')))
    return {k: d[k] for k in ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'out_w', 'out_b')}


def test_demo_folder_end_to_end(tmp_path):
    model = body_model()
    out = batch.fit_folder(model, os.path.join(DATA, 'keypoints'), os.path.join(DATA, '3DOH50K_Parameters.txt'),
                           str(tmp_path / 'results'), vposer=_vposer(), image_height=1536.0, save_meshes=True)
    r = out['0000']
    assert r['frames'] == ['00001'] and r['params'].shape == (1, 118)
    # initial guess: translation and scale are those of the reference's init_guess (the orientation depends on the SVD's
    # sign convention, tests/test_gpu_init_guess.py); zero shape and embedding (fix_params)
    x0 = r['init'][0]
    ref0 = G['x0']                                            # betas(10) go(3) transl(3) scale(1) embedding(32)
    assert np.all(x0[:10] == 0) and np.all(x0[86:] == 0) and np.all(x0[13:82] == 0)
    assert abs(x0[85] - ref0[16]) < 1e-4 * abs(ref0[16])
    # the fit: inside the spread of the reference's own float32 fits of this frame (1e-6-perturbed starts)
    spread = G['fit_spread32']
    assert np.isfinite(r['final_loss'][0]) and r['final_loss'][0] <= 1.05 * spread.max(), (r['final_loss'], spread)
    # result file: the reference's layout and keys (utils.py:744-766, 859-864), feet / hands zeroed, pose = go | body_pose
    path = tmp_path / 'results' / '0000' / '00001' / '000.pkl'
    assert str(path) == r['files'][0] and path.exists()
    with open(path, 'rb') as f:
        res = pickle.load(f)
    assert set(res) == {'betas', 'global_orient', 'transl', 'scale', 'loss', 'pose_embedding', 'body_pose', 'pose'}
    assert res['betas'].shape == (1, 10) and res['pose'].shape == (1, 72) and res['body_pose'].shape == (1, 69)
    bp = res['body_pose'][0]
    assert np.all(bp[18:24] == 0) and np.all(bp[27:33] == 0) and np.all(bp[57:] == 0) and np.any(bp[:18] != 0)
    assert np.array_equal(res['pose'][0, :3], res['global_orient'][0]) and np.array_equal(res['pose'][0, 3:], bp)
    assert np.array_equal(res['pose_embedding'][0], r['params'][0, 86:118])
    # mesh of the saved parameters
    obj = tmp_path / 'results' / 'meshes' / '0000' / '00001' / '000.obj'
    lines = obj.read_text().splitlines()
    assert sum(l.startswith('v ') for l in lines) == 6890 and sum(l.startswith('f ') for l in lines) == model['faces'].shape[0]


def test_sequence_folder(tmp_path):
    """Three frames (the demo frame under three names), one camera file missing in the last, fitted as a sequence: one
    result file per frame.  The demo frame ends at a loss of ~37 k, above the reference's restart threshold of 5000
    (main.py:76-79 / init_guess.py:137-146), so every frame is fitted from its own initial guess like the reference would:
    frame 1 repeats frame 0 bit for bit, frame 2 (five views) differs.  The warm-started chain itself is
    tests/test_gpu_sequence.py."""
    root = tmp_path / 'keypoints' / 'walk'
    for v in range(6):
        (root / ('Camera%02d' % v)).mkdir(parents=True)
        for i, fn in enumerate(('00001', '00002', '00003')):
            if v == 5 and i == 2:
                continue
            shutil.copy(os.path.join(DATA, 'keypoints', '0000', 'Camera%02d' % v, '00001_keypoints.json'),
                        root / ('Camera%02d' % v) / (fn + '_keypoints.json'))
    out = batch.fit_folder(body_model(), str(tmp_path / 'keypoints'), os.path.join(DATA, '3DOH50K_Parameters.txt'),
                           str(tmp_path / 'res'), vposer=_vposer(), is_seq=True)
    r = out['walk']
    assert r['frames'] == ['00001', '00002', '00003'] and all(os.path.exists(p) for p in r['files'])
    assert np.all(np.isfinite(r['final_loss']))
    assert np.all(r['restarted'])
    assert np.array_equal(r['params'][1], r['params'][0]) and r['n_closure'][1] == r['n_closure'][0]
    assert not np.array_equal(r['params'][2], r['params'][0])


def test_synthetic_folder_with_3d_annotation(tmp_path):
    """A synthetic serial written in the reference's file formats (camera text file, keypoint json with
    'pose_keypoints_2d' and 'pose_keypoints_3d'): 3 frames x 4 views, fitted with use_3d - the 3-D targets drive the
    initial alignment (init_guess.py:84-85) and the D3 term - and as a sequence."""
    import json
    from mvsmplfitting_amd import synthetic as syn
    from tests.gpu_helpers import make_engine
    model = body_model()
    V, F = 4, 3
    cams = syn.make_camera_ring(V)
    eng = make_engine(model)
    fr = syn.make_frames(F, seed0=50)
    xgt = np.zeros((F, 118), np.float32)
    for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85), scale=(85, 86)).items():
        xgt[:, a:b] = fr[k]
    eng.set_problems(cams, np.zeros((F, V, 17, 2), np.float32), np.ones((F, V, 17), np.float32))
    _, joints = eng.vertices(xgt)
    joints = joints.cpu().numpy()
    gt, conf = syn.make_observations(joints, cams, seed=5)
    # the camera file: three 3-number rows of K and three 4-number rows of [R|t] per camera (utils.py:352-394)
    cam_R, cam_t, cam_f, cam_c = (np.asarray(a, np.float64) for a in cams)
    with open(tmp_path / 'cams.txt', 'w') as fh:
        for v in range(V):
            fh.write('%d\n' % v)
            K = np.array([[cam_f[v], 0, cam_c[v, 0]], [0, cam_f[v], cam_c[v, 1]], [0, 0, 1]])
            for r in K:
                fh.write(' '.join('%.10f' % x for x in r) + '\n')
            fh.write('0 0\n')
            for r in np.hstack([cam_R[v], cam_t[v][:, None]]):
                fh.write(' '.join('%.10f' % x for x in r) + '\n')
    for v in range(V):
        d = tmp_path / 'keypoints' / 's0' / ('Camera%02d' % v)
        d.mkdir(parents=True)
        for f in range(F):
            k2 = np.concatenate([gt[f, v], conf[f, v][:, None]], 1).reshape(-1)
            k3 = np.concatenate([joints[f], np.ones((17, 1), np.float32)], 1).reshape(-1)
            with open(d / ('%05d_keypoints.json' % f), 'w') as fh:
                json.dump(dict(version=1.0, people=[dict(pose_keypoints_2d=[float(x) for x in k2],
                                                         pose_keypoints_3d=[float(x) for x in k3])]), fh)
    for seq in (False, True):
        out = batch.fit_folder(model, str(tmp_path / 'keypoints'), str(tmp_path / 'cams.txt'), str(tmp_path / ('res%d' % seq)),
                               use_3d=True, is_seq=seq, engine=eng)['s0']
        assert out['frames'] == ['%05d' % f for f in range(F)] and all(os.path.exists(p) for p in out['files'])
        # with exact 3-D targets the alignment starts next to the truth and the fit recovers the root translation
        assert np.abs(out['init'][:, 82:85] - xgt[:, 82:85]).max() < 0.25
        assert np.abs(out['params'][:, 82:85] - xgt[:, 82:85]).max() < 0.05
        assert np.all(np.isfinite(out['final_loss']))
    eng.close()
