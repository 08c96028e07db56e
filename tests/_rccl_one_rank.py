"""Child process of test_allreduce_flat_through_a_one_rank_rccl_communicator: a one-rank RCCL communicator made with ctypes,
cara_allreduce_flat on it.  Exit code 77 = nothing to test here (no librccl / no communicator)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cara_amd import _lib as L  # noqa: E402


def main():
    if not torch.cuda.is_available():
        print("no GPU")
        return 77
    rccl = None
    for name in ("librccl.so.1", "librccl.so"):
        try:
            rccl = C.CDLL(name)
            break
        except OSError:
            pass
    if rccl is None:
        print("no librccl")
        return 77

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")   # (a HIP context before RCCL asks for one)
    uid = UniqueId()
    if rccl.ncclGetUniqueId(C.byref(uid)) != 0:
        print("ncclGetUniqueId failed")
        return 77
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    if rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) != 0:
        print("ncclCommInitRank failed")
        return 77
    lib = L.lib()
    g = torch.Generator(device="cpu").manual_seed(3)
    buf = torch.randn(121923 + 1, generator=g).cuda()
    ref = buf.clone()
    lib.cara_allreduce_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    rc = lib.cara_allreduce_flat(comm, L.ptr(buf), buf.numel(), L.stream())
    torch.cuda.synchronize()
    ok = rc == 0 and torch.equal(buf, ref)
    ok = ok and lib.cara_allreduce_flat(None, L.ptr(buf), buf.numel(), L.stream()) != 0
    ok = ok and lib.cara_allreduce_flat(comm, None, buf.numel(), L.stream()) != 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    print("allreduce ok" if ok else f"allreduce WRONG (rc {rc})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
