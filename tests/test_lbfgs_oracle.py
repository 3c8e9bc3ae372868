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


@pytest.mark.parametrize('kind,D', [('quad', 49), ('quad', 86), ('gmof', 49), ('gmof', 86)])
def test_gtd_exit_after_direction_follows_reference(kind, D):
    """`gtd > -tolerance_change` right after a direction computation (lbfgs_ls.py:379-380) ends step() without a new
    closure call; run_fitting's gtol test (fitting.py:115-116) then reads the gradient the LAST closure left - not
    zeros - and the outer loop keeps stepping until ftol stops it.  Goldens: the reference's LBFGS class with
    tolerance_grad = 1e-12 and a large tolerance_change (oracle/make_golden.py:GTD_CASES)."""
    key = '%s_%d_gtd' % (kind, D)
    fn, x0 = ln.kat_objective(kind, D)
    opt = ln.LbfgsOracle(x0, fn, tol_grad=1e-12, tol_change=float(KAT[key + '_tc']))
    prev, losses = ln.run_fitting(opt, segments=[(0, 10), (10, 13), (13, D)])
    ref_trace = KAT[key + '_trace']
    assert len(opt.trace) == int(KAT[key + '_n'])
    for i in range(len(ref_trace)):
        assert np.abs(opt.trace[i][0] - ref_trace[i][:D]).max() < 1e-8, (key, i)
    assert abs(prev - float(KAT[key + '_final'])) < 1e-10 and np.abs(opt.x - KAT[key + '_xf']).max() < 1e-10
    gtd = [e for e in opt.exits if e[0] == 'gtd']
    assert len(gtd) >= 2 and all(n_iter > 1 for _, n_iter in gtd)          # after a direction, and more than once
    assert len(losses) > opt.exits.index(gtd[0]) + 1                       # the outer loop went on after the first one
