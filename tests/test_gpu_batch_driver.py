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
