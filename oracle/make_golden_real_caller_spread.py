"""TEST INFRASTRUCTURE - the yard-stick of tests/test_gpu_real_caller.py's VPoser case with all four yaml stages: how far the
REFERENCE's own float32 fit of that synthetic frame moves when the start is perturbed in the last bits.  The unmodified
non_linear_solver (code/utils/non_linear_solver.py:37-288) is run un-patched from N starts = the rest pose + N - 1 copies whose
non-zero start values (scale = 1) are perturbed by 1e-6 (relative) and whose zero entries get 1e-7 absolute noise.

    python -m oracle.make_golden_real_caller_spread [n]      (build container; ~10 s per fit)

Writes tests/golden/real_caller_vposer_spread.npz: final32 [n] (float32 fits), final64 (the float64 fit of the unperturbed start)."""
from __future__ import annotations

import os
import sys

import numpy as np

from oracle import ref_import as ri
from oracle.make_golden import GOLD


def main(n=24):
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.test_real_caller import _problem
    from tests.test_gpu_real_caller import _run, _setting_and_data
    ri.load()
    from utils import non_linear_solver as nls
    prob = _problem(True)
    rng = np.random.default_rng(0)
    finals = []
    for i in range(n):
        rp, _, _ = _setting_and_data(*prob, True, 'float32')
        x0 = rp.get_flat().astype(np.float64)
        if i:
            x0 = x0 * (1.0 + 1e-6 * rng.standard_normal(x0.shape)) + 1e-7 * rng.standard_normal(x0.shape)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            res, _ = _run(nls, prob, True, 'lbfgsls', 'float32', 4, start=None if i == 0 else x0)
        finals.append(float(res['loss']))
        print(i, finals[-1], flush=True)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        res64, _ = _run(nls, prob, True, 'lbfgsls', 'float64', 4)
    np.savez_compressed(os.path.join(GOLD, 'real_caller_vposer_spread.npz'), final32=np.asarray(finals), final64=np.array(float(res64['loss'])))
    print('reference float32, %d starts: %s; float64 %.4f' % (n, np.sort(np.asarray(finals)), float(res64['loss'])))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
