"""CPU: the batch driver's folder parsing against what the reference's own FittingData / load_camera_para produced for
its shipped demo (tests/golden/demo_fit_smpl.npz was written from the reference's parsers by oracle/make_golden_demo.py;
tests/golden/demo_data holds the same input files)."""
import os

import numpy as np

from mvsmplfitting_amd import batch
from mvsmplfitting_amd import io_formats as iof
from tests.helpers import GOLD

DATA = os.path.join(GOLD, 'demo_data')


def test_folder_inputs_equal_the_reference_demo():
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    serials = batch.list_frames(os.path.join(DATA, 'keypoints'))
    assert [(s, c, [f for f, _ in fr]) for s, c, fr in serials] == [('0000', ['Camera%02d' % v for v in range(6)], ['00001'])]
    kp = batch.load_serial(serials[0][2], 6)
    assert np.array_equal(kp[0], g['keypoints'].reshape(6, 17, 3))
    extris, intris = iof.load_camera_para(os.path.join(DATA, '3DOH50K_Parameters.txt'))
    assert np.array_equal(extris, g['extris']) and np.array_equal(intris, g['intris'])
    # the problem tensors the driver hands to the engine = the reference's camera / target tensors of the demo
    # (model_type 'smpllsp' -> pose_format 'lsp14' with use_hip: all 17 joint weights are 1, init.py:63-69)
    assert np.array_equal(kp[0, :, :, :2], g['gt_xy'].reshape(6, 17, 2))
    assert np.array_equal(kp[0, :, :, 2], g['conf'].reshape(6, 17))
    assert np.allclose(extris[:, :3, :3], g['cam_R']) and np.allclose(extris[:, :3, 3], g['cam_t'])
    assert np.allclose(intris[:, 0, 0], g['cam_f']) and np.allclose(intris[:, :2, 2], g['cam_c'])


def test_missing_view_and_ragged_frames(tmp_path):
    """A frame that one camera has no file for, and a camera folder with an extra frame."""
    import shutil
    root = tmp_path / 'keypoints' / 'seq'
    src = os.path.join(DATA, 'keypoints', '0000')
    for v in range(3):
        (root / ('Camera%02d' % v)).mkdir(parents=True)
        for fn in ('00001', '00002'):
            if v == 1 and fn == '00002':
                continue
            shutil.copy(os.path.join(src, 'Camera%02d' % v, '00001_keypoints.json'), root / ('Camera%02d' % v) / (fn + '_keypoints.json'))
    (serial, cams, frames), = batch.list_frames(str(tmp_path / 'keypoints'))
    assert serial == 'seq' and len(cams) == 3 and [f for f, _ in frames] == ['00001', '00002']
    assert frames[1][1][1] is None and frames[1][1][0] is not None
    kp = batch.load_serial(frames, 3)
    assert kp.shape == (2, 3, 17, 3) and np.all(kp[1, 1] == 0) and np.array_equal(kp[1, 0], kp[0, 0])


def test_fit_folder_host_logic_on_the_demo_folder(tmp_path):
    """fit_folder end to end on CPU: the engine is the recording stub that answers with the float64 oracle (tests only; there
    is no GPU in the build container), everything else - folder walk, initial guess plumbing, stage weights, result
    dictionaries, the reference's file layout - is the shipped code.  The fitted parameters equal the reference-recorded
    demo fit's quality (its float64 run ends at fit_final64)."""
    import pickle
    from tests.helpers import body_model
    from tests.stub_engine import StubMvFit
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    d = dict(np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')))
    vpw = {k: d[k] for k in ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'out_w', 'out_b')}
    model = body_model()
    eng = StubMvFit(model, vposer=vpw)
    out = batch.fit_folder(model, os.path.join(DATA, 'keypoints'), os.path.join(DATA, '3DOH50K_Parameters.txt'),
                           str(tmp_path / 'results'), vposer=vpw, image_height=1536.0, engine=eng)['0000']
    # the initial guess equals the reference's own (same LAPACK behind the restatement and the reference): x0 of the golden
    ref0 = g['x0']
    x0 = out['init'][0]
    assert np.abs(x0[10:13] - ref0[10:13]).max() < 1e-5 and np.abs(x0[82:85] - ref0[13:16]).max() < 1e-4
    assert abs(x0[85] - ref0[16]) < 1e-5 * abs(ref0[16])
    # float64 oracle optimiser from the same start on the same objective: the reference's float64 fit
    assert abs(out['final_loss'][0] - float(g['fit_final64'])) <= 0.02 * float(g['fit_final64'])
    with open(out['files'][0], 'rb') as f:
        res = pickle.load(f)
    assert out['files'][0] == str(tmp_path / 'results' / '0000' / '00001' / '000.pkl')
    assert set(res) == {'betas', 'global_orient', 'transl', 'scale', 'loss', 'pose_embedding', 'body_pose', 'pose'}
    bp = res['body_pose'][0]
    assert np.all(bp[18:24] == 0) and np.all(bp[27:33] == 0) and np.all(bp[57:] == 0) and np.any(bp[:18] != 0)


def test_more_camera_folders_than_cameras_is_an_error(tmp_path):
    import shutil
    import pytest
    from tests.helpers import body_model
    from tests.stub_engine import StubMvFit
    root = tmp_path / 'keypoints' / 's'
    for v in range(7):
        (root / ('Camera%02d' % v)).mkdir(parents=True)
        shutil.copy(os.path.join(DATA, 'keypoints', '0000', 'Camera00', '00001_keypoints.json'), root / ('Camera%02d' % v) / '00001_keypoints.json')
    model = body_model()
    with pytest.raises(ValueError):
        batch.fit_folder(model, str(tmp_path / 'keypoints'), os.path.join(DATA, '3DOH50K_Parameters.txt'), str(tmp_path / 'r'),
                         engine=StubMvFit(model))


def test_initial_guess_uses_only_the_views_a_frame_has():
    """main.py:44-66 drops the views without annotation before init_guess: a frame with two of three views is
    triangulated from those two (not from three with a 1e-6-weighted ray through pixel (0, 0)), a frame with ONE view
    takes the single-view depth guess with that view's camera - per frame, whatever the rig's camera count."""
    from mvsmplfitting_amd import init_guess as ig
    from oracle import init_guess_np as ign
    from oracle import triangulate_np as tn
    from oracle import umeyama_np as un
    from tests.helpers import body_model
    from tests.stub_engine import StubMvFit
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    model = body_model()
    eng = StubMvFit(model)
    kp6 = g['keypoints'].reshape(6, 17, 3).astype(np.float32)
    V = 3
    kp = np.stack([kp6[:V]] * 3)                                             # three frames of a 3-camera rig
    mask = np.array([[1, 1, 1], [1, 0, 1], [0, 0, 1]], bool)
    kp[~mask] = 0.0                                                          # what batch.load_serial leaves for a missing file
    ex, it = g['extris'][:V], g['intris'][:V]
    eng.set_problems(tuple(g[k][:V] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c')), kp[..., :2], kp[..., 2])
    out = ig.init_guess_batch(eng, ex, it, kp, view_mask=mask)
    rest = ig.rest_keypoints(eng).numpy()
    want = [tn.recompute3d(ex, it, kp[0]), tn.recompute3d(ex[[0, 2]], it[[0, 2]], kp[1][[0, 2]]),
            ign.single_view_joints3d(rest, ex[2], it[2], kp[2, 2])]
    for f in range(3):
        assert np.abs(out['joints3d'][f].numpy() - want[f]).max() < 1e-9 * np.abs(want[f]).max(), f
        r, t, s_, _ = un.umeyama(rest[[5, 6, 11, 12]], want[f][[5, 6, 11, 12]], True)
        assert np.abs(out['transl'][f].numpy() - t).max() < 1e-8 and abs(float(out['scale'][f]) - s_) < 1e-9
    # without the mask the absent views would take part (the old behaviour): a different - wrong - point
    bad = ig.init_guess_batch(eng, ex, it, kp)['joints3d'][1].numpy()
    assert np.abs(bad - want[1]).max() > 1e-6
