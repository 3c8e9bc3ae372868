"""GPU: mvfit_full_pose against the reference's own ModelOutput.full_pose (SMPL.forward, body_models_scale.py:392-412)
with the body pose from VPoser.decode(z, 'aa') - tests/golden/full_pose.npz, written by oracle/make_golden_full_pose.py
from the shipped checkpoint's decoder and from a decoder that reaches all four matrix -> quaternion branches."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn
from tests.gpu_helpers import make_engine, to118
from tests.helpers import GOLD, body_model

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(GOLD, 'full_pose.npz')))


def _decoder(name):
    if name == 'wild':
        return syn.make_vposer_decoder(seed=2, gain=1.0, identity_bias=False)
    d = dict(np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')))
    return {k: d[k] for k in ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'out_w', 'out_b')}


@pytest.mark.parametrize('name', ['real', 'wild'])
def test_full_pose_with_vposer(name):
    xs = G[name + '_x']
    B = xs.shape[0]
    eng = make_engine(body_model(0, 4), _decoder(name))
    cams = syn.make_camera_ring(2)
    eng.set_problems(cams, np.zeros((B, 2, 17, 2), np.float32), np.ones((B, 2, 17), np.float32))
    x = np.stack([to118(r, True) for r in xs]).astype(np.float32)
    fp = eng.full_pose(x, flags=_lib.F_VPOSER).cpu().numpy()
    ref = G[name + '_full_pose']
    assert np.array_equal(fp[:, :3], x[:, 10:13])
    # axis-angle of the decoded rotations, float32 decoder on the device vs the reference in float64
    assert np.abs(fp - ref).max() < 2e-5, np.abs(fp - ref).max()
    eng.close()


def test_full_pose_without_vposer_is_the_parameters():
    eng = make_engine(body_model(0, 4))
    cams = syn.make_camera_ring(2)
    B = 3
    eng.set_problems(cams, np.zeros((B, 2, 17, 2), np.float32), np.ones((B, 2, 17), np.float32))
    x = np.zeros((B, 118), np.float32)
    x[:, :86] = np.random.default_rng(3).normal(0, 0.3, (B, 86)).astype(np.float32)
    fp = eng.full_pose(x).cpu().numpy()
    assert np.array_equal(fp, x[:, 10:82])
    eng.close()
