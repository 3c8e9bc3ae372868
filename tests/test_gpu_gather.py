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


@pytest.mark.timeout(600)
def test_gather_over_a_two_rank_rccl_communicator():
    """mvfit_gather between two GPUs over RCCL / xGMI: two processes (tests/gather_2rank_worker.py), one GPU each, a raw
    ncclComm_t made from a shared ncclUniqueId; every rank must end with both ranks' rows.  Skipped on a one-GPU box (the
    build sessions' and the round-end test box): it runs the day the suite sees two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL refuses two ranks on one device)')
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29751', os.path.join(root, 'tests', 'gather_2rank_worker.py')],
                       cwd=root, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0 and r.stdout.count('gather ok') == 2, r.stdout[-2000:] + r.stderr[-4000:]
