"""Worker of tests/test_generate_cpu.py::test_bench_generation_leg_two_ranks_gloo: one rank of bench.generation_leg on CPU
(backend gloo, tiny synthetic model) - the collective sequence of the N > 1 bench line, with and without a failure injected
on one rank (UCE_BENCH_FAIL_RANK)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uce_amd import edit as E  # noqa: E402
from uce_amd.sd import pipeline as sdp  # noqa: E402


def main(out_dir: str) -> None:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    try:
        pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=False)
        slab = E.WeightSlab.from_modules(E.collect_uce_modules(pipe.unet), "cpu").data
        blob = slab * 1.25 if rank == 0 else torch.zeros_like(slab)      # only rank 0 holds the edited weights
        res = bench.generation_leg("cpu", world, 2, 2, blob, batch=2, model_id="tiny-sd-test", dtype=torch.float32, vae=False)
        with open(os.path.join(out_dir, f"gen_w{world}_r{rank}.json"), "w") as fh:
            json.dump(res, fh)
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
