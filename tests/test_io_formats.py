"""CPU: the host-side file formats either side of the path (mvsmplfitting_amd/io_formats.py) - round trips, and
against the reference's own parsers when its tree is mounted."""
import json
import os
import pickle

import numpy as np
import pytest

from mvsmplfitting_amd import io_formats as iof
from oracle import ref_import as ri


def _write_fixture(tmp_path, V=3):
    rng = np.random.default_rng(0)
    cam = tmp_path / 'camparams.txt'
    lines = []
    Ks, Es = [], []
    for v in range(V):
        K = np.array([[2400.0 + v, 0, 1024.5], [0, 2399.0 - v, 768.25], [0, 0, 1]])
        E = np.hstack([np.linalg.qr(rng.normal(size=(3, 3)))[0], rng.normal(size=(3, 1))])
        Ks.append(K); Es.append(E)
        lines.append(str(v))                              # a camera index line (1 word: ignored by the parser)
        for r in K: lines.append(' '.join(repr(float(x)) for x in r))
        lines.append('0 0')                               # distortion line (2 words: ignored)
        for r in E: lines.append(' '.join(repr(float(x)) for x in r))
    cam.write_text('\n'.join(lines) + '\n')
    kp = rng.uniform(0, 2000, (2, 17, 3)).astype(np.float32)
    kj = tmp_path / 'kp.json'
    kj.write_text(json.dumps({'version': 1.1, 'people': [{'pose_keypoints_2d': kp[p].flatten().tolist()} for p in range(2)]}))
    return cam, np.array(Ks), np.array(Es), kj, kp


def test_camera_and_keypoint_files(tmp_path):
    cam, Ks, Es, kj, kp = _write_fixture(tmp_path)
    extris, intris = iof.load_camera_para(str(cam))
    assert extris.shape == (3, 4, 4) and intris.shape == (3, 3, 3)
    assert np.array_equal(intris, Ks) and np.array_equal(extris[:, :3], Es) and np.array_equal(extris[:, 3], np.tile([0, 0, 0, 1.0], (3, 1)))
    people = iof.read_keypoints(str(kj))
    assert len(people) == 2 and people[0].dtype == np.float32 and np.array_equal(people[1], kp[1])
    cams, gt, conf = iof.problem_tensors(extris, intris, [people[0]] * 3)
    assert gt.shape == (1, 3, 17, 2) and conf.shape == (1, 3, 17) and cams[2].shape == (3,) and cams[3].shape == (3, 2)
    assert np.allclose(cams[2], Ks[:, 0, 0]) and np.allclose(cams[3], Ks[:, :2, 2])


def test_result_files(tmp_path):
    x = np.arange(118, dtype=np.float32) * 0.01
    r = iof.result_dict(x, loss=12.5)
    want = x[13:82].copy()                       # non-VPoser branch zeroes feet / hands too (utils/utils.py:761-766)
    want[18:24] = 0; want[27:33] = 0; want[57:] = 0
    assert r['pose'].shape == (1, 72) and np.array_equal(r['pose'][0, 3:], want) and r['loss'] == 12.5
    assert np.array_equal(r['body_pose'][0], want) and np.array_equal(r['pose'][0, :3], x[10:13])
    bp = np.ones(69, np.float32)
    rv = iof.result_dict(x, body_pose_decoded=bp)
    z = rv['body_pose'][0]
    assert np.all(z[18:24] == 0) and np.all(z[27:33] == 0) and np.all(z[57:] == 0) and np.all(z[:18] == 1) and np.all(z[33:57] == 1)
    assert np.array_equal(rv['pose_embedding'][0], x[86:118])
    path = iof.save_result_pkl(str(tmp_path), '0000', '00001', rv)
    assert path.endswith(os.path.join('0000', '00001', '000.pkl'))
    with open(path, 'rb') as f:
        back = pickle.load(f)
    assert np.array_equal(back['pose'], rv['pose'])
    obj = tmp_path / 'm.obj'
    iof.save_obj(str(obj), np.eye(3), np.array([[0, 1, 2]]))
    txt = obj.read_text().splitlines()
    assert txt[0].startswith('v 1.000000 0.000000') and txt[-1] == 'f 1 2 3'


@pytest.mark.skipif(not ri.available(), reason='reference tree not mounted')
def test_parsers_equal_the_reference_ones(tmp_path):
    cam, Ks, Es, kj, kp = _write_fixture(tmp_path, V=4)
    ref = ri.load()
    e_ref, i_ref = ref.utils.load_camera_para(str(cam))
    e, i = iof.load_camera_para(str(cam))
    assert np.array_equal(e, e_ref) and np.array_equal(i, i_ref)
    import importlib
    dp = importlib.import_module('utils.data_parser')
    k_ref = dp.read_keypoints(str(kj), use_hands=False, use_face=False)
    mine = iof.read_keypoints(str(kj))
    assert len(mine) == len(k_ref.keypoints) and all(np.array_equal(a, b) for a, b in zip(mine, k_ref.keypoints))
