"""Worker of tests/test_gpu_gather.py::test_gather_over_a_two_rank_rccl_communicator: one process per GPU (launched by
torch.distributed.run), gloo only to hand rank 0's ncclUniqueId to the other rank; the data path is mvfit_gather on a raw
ncclComm_t (RCCL over xGMI)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    dist.init_process_group('gloo')
    torch.cuda.set_device(local)
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = _UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    box = [bytes(bytearray(C.string_at(C.addressof(uid), 128)))]
    dist.broadcast_object_list(box, src=0)
    C.memmove(C.addressof(uid), box[0], 128)
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    from mvsmplfitting_amd import synthetic as syn
    from mvsmplfitting_amd.engine import MvFit
    eng = MvFit(syn.make_body_model(0, skin_topk=4), device=local)
    rows = 16
    x = (torch.arange(rows * 120, dtype=torch.float32, device='cuda').reshape(rows, 120) + 10000.0 * rank)
    out = eng.gather(comm, x, world)
    eng.sync()
    for r in range(world):
        want = torch.arange(rows * 120, dtype=torch.float32, device='cuda').reshape(rows, 120) + 10000.0 * r
        assert torch.equal(out[r], want), (rank, r)
    print('gather ok rank %d of %d' % (rank, world), flush=True)
    eng.close()
    rccl.ncclCommDestroy(comm)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
