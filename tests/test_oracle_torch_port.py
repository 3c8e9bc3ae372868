"""CPU: the PyTorch-CPU port used as bench.py's cpu_baseline computes the reference closure."""
import numpy as np
import pytest
import torch

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import closure_torch as ct
from tests.helpers import CASES, load_case, oracle_for


@pytest.mark.parametrize('name', ['l2_s3_v6', 'vpwild_s2_v8', 'l2_angle_drop_v8'])
def test_torch_port_matches_golden(name):
    cfg, g, model, vpw, gmm, wts, cams = load_case(name)
    for b in (0, 2):
        tc = ct.TorchClosure(model, cams, g['gt_xy'][b], g['conf'][b], dtype=torch.float64, vposer=vpw)
        L, grad = tc.evaluate(g['x'][b], wts, cfg['use_vposer'])
        assert abs(L - g['loss64'][b]) <= 1e-11 * abs(g['loss64'][b])
        assert np.abs(grad - g['grad64'][b]).max() <= 1e-9 * np.abs(g['grad64'][b]).max()
