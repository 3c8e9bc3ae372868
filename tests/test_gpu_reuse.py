"""GPU: the opt-in MVFIT_F_REUSE_OUTER_VALUE (include/mvfit.h) - LBFGS.step() opens with a closure call
(lbfgs_ls.py:279-283) at the point the previous step() of the same stage ended on; with the flag the device optimiser
feeds the loss / gradient it still holds instead of evaluating again.  It must not change a single iterate: parameters
and final losses bit-identical to the default fit, only the closure count drops (by the number of skipped calls)."""
import numpy as np
import pytest

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd.engine import stage_weights
from tests.test_gpu_async import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('sparse', [False, True])
def test_reuse_of_the_step_start_value_changes_no_iterate(sparse):
    eng, x0 = _setup(B=7)
    base = _lib.F_SPARSE_VERTS if sparse else 0
    xa, sa = eng.fit(x0, stage_weights(1536.0, flags=base))
    xb, sb = eng.fit(x0, stage_weights(1536.0, flags=base | _lib.F_REUSE_OUTER_VALUE))
    assert np.array_equal(xa.cpu().numpy(), xb.cpu().numpy())
    assert np.array_equal(sa['final_loss'].cpu().numpy(), sb['final_loss'].cpu().numpy())
    assert np.array_equal(sa['n_iter'].cpu().numpy(), sb['n_iter'].cpu().numpy())
    na, nb = sa['n_closure'].cpu().numpy(), sb['n_closure'].cpu().numpy()
    assert np.all(nb < na) and np.all(nb > 0.8 * na), (na, nb)          # 8-10 % of the calls are step-start re-evaluations
    eng.close()
