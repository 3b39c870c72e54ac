"""Worker of tests/test_generate_gpu.py: one rank, backend nccl (= RCCL on ROCm), on cuda:0 - the edited-weight
broadcast of generate.broadcast_uce_weights through a real RCCL communicator (the 8-GPU path is the same call with
world_size 8; a single-GPU box can only host one rank per device)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import generate  # noqa: E402


def main(path: str) -> None:
    from safetensors.torch import load_file
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['UCE_TEST_PORT']}", rank=0, world_size=1,
                            device_id=dev)
    try:
        got = generate.broadcast_uce_weights(path, None, dev, 0, 1, force_collective=True)
        want = load_file(path)
        assert list(got) == list(want)
        for k in want:
            assert got[k].is_cuda and torch.equal(got[k].cpu(), want[k]), k
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                        # the counters' reduction of generate_images
        torch.cuda.synchronize()
        assert bool((t == 1).all())
        # the same collective through the C ABI (uce_bcast: what a host without torch calls): a single-rank communicator
        # created with torch's own RCCL, made globally visible so that the library resolves ncclBroadcast from that copy
        import ctypes
        from uce_amd import lib as L, edit as E
        rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]

        uid, comm = UniqueId(), ctypes.c_void_p()
        rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
        H = E.UceHandle.get(dev)
        blob = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        keep = blob.clone()
        rc = H.lib.uce_bcast(H._h, blob.data_ptr(), blob.numel() * 4, 0, comm, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, L.load().uce_strerror(rc)
        assert torch.equal(blob, keep)
        assert H.lib.uce_bcast(H._h, blob.data_ptr(), 16, 0, None, None) == L.EINVAL       # no communicator
        rccl.ncclCommDestroy(comm)
        print("RCCL_BCAST_OK", dist.get_backend(), len(want), "uce_bcast ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
