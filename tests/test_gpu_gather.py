"""GPU: mvfit_gather - the all-gather of the fitted parameters for hosts that own a raw RCCL communicator.  One MI355X
here, so the communicator has one rank (the multi-rank data path is the same ncclAllGather call; the sharded fit itself
is covered by tests/test_sharding_gloo.py and tests/test_gpu_sharded_fit.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mvsmplfitting_amd.engine import MvFitError
from tests.gpu_helpers import make_engine
from tests.helpers import body_model

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):          # ncclUniqueId: 128 opaque bytes, passed by value
    _fields_ = [('internal', C.c_char * 128)]


def _rccl():
    """The RCCL copy of this process: the one PyTorch ships (what a torch host would have created its communicator with)."""
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    if not os.path.exists(path):
        pytest.skip('no librccl.so next to torch')
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


@pytest.mark.timeout(180)
def test_gather_over_a_raw_communicator():
    eng = make_engine(body_model())
    rccl = _rccl()
    uid = _UniqueId()
    comm = C.c_void_p()
    if rccl.ncclGetUniqueId(C.byref(uid)) != 0 or rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) != 0:
        pytest.skip('RCCL could not create a one-rank communicator on this box')
    try:
        x = torch.arange(32 * 120, dtype=torch.float32, device='cuda').reshape(32, 120) * 0.25
        out = eng.gather(comm, x, 1)
        eng.sync()
        assert out.shape == (1, 32, 120) and torch.equal(out[0], x)
        # nothing to move: accepted, no call into RCCL
        assert eng._lib.mvfit_gather(eng._ctx, comm, x.data_ptr(), out.data_ptr(), 0) == 0
    finally:
        rccl.ncclCommDestroy(comm)
    with pytest.raises(MvFitError):
        eng.gather(None, x, 1)              # null communicator: MVFIT_E_ARG with a message, never a crash
    eng.close()
