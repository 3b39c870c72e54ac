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
        print("RCCL_BCAST_OK", dist.get_backend(), len(want))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
