"""CPU: NumPy L-BFGS / strong-Wolfe / run_fitting restatement vs reference trajectories."""
import os

import numpy as np
import pytest

from oracle import lbfgs_np as ln
from tests.helpers import GOLD

KAT = dict(np.load(os.path.join(GOLD, 'lbfgs_kat.npz')))


@pytest.mark.parametrize('kind', ['quad', 'rosen', 'gmof'])
@pytest.mark.parametrize('D', [49, 86])
def test_lbfgs_oracle_follows_reference(kind, D):
    key = '%s_%d' % (kind, D)
    fn, x0 = ln.kat_objective(kind, D)
    opt = ln.LbfgsOracle(x0, fn)
    prev, _ = ln.run_fitting(opt, segments=[(0, 10), (10, 13), (13, D)])
    ref_trace = KAT[key + '_trace']
    n = min(len(ref_trace), len(opt.trace), 40)
    # step-for-step over the first closures (rounding amplification afterwards on 'rosen')
    for i in range(n):
        assert np.abs(opt.trace[i][0] - ref_trace[i][:D]).max() < 1e-8, (key, i)
        assert abs(opt.trace[i][1] - ref_trace[i][D]) <= 1e-8 * max(1.0, abs(ref_trace[i][D]))
    if kind != 'rosen':
        assert len(opt.trace) == int(KAT[key + '_n'])
        assert np.abs(opt.x - KAT[key + '_xf']).max() < 1e-10
    else:
        assert abs(len(opt.trace) - int(KAT[key + '_n'])) <= 10
        assert np.abs(opt.x - KAT[key + '_xf']).max() < 1e-4
    assert abs(prev - float(KAT[key + '_final'])) < 1e-8


def test_cubic_interpolate_cases():
    # d2_square < 0 -> midpoint of bounds
    assert ln.cubic_interpolate(0.0, 0.0, 1.0, 1.0, 2.0 / 3.0, 1.0, bounds=(0.2, 0.6)) == pytest.approx(0.4)
    # symmetric parabola f = (x-0.5)^2: minimum at 0.5
    assert ln.cubic_interpolate(0.0, 0.25, -1.0, 1.0, 0.25, 1.0) == pytest.approx(0.5)
    assert ln.cubic_interpolate(1.0, 0.25, 1.0, 0.0, 0.25, -1.0) == pytest.approx(0.5)
