"""Worker of tests/test_generate_cpu.py::test_eight_rank_gloo_generation_covers_the_reference_file_list: one rank of
generate.generate_images on CPU (backend gloo, tiny synthetic model, PNG output) over the committed coco_30k rows; writes its
own stats next to the images."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import generate  # noqa: E402


def main(out_dir: str, csv_path: str) -> None:
    rank = int(os.environ["RANK"])
    torch.set_num_threads(1)
    stats = generate.generate_images("tiny-sd-test", None, csv_path, out_dir, exp_name="coco", device="cpu", torch_dtype=torch.float32,
                                     guidance_scale=7.5, num_inference_steps=1, num_images_per_prompt=1, synthetic=True,
                                     batch_prompts=3, png_workers=1)
    with open(os.path.join(out_dir, f"stats_r{rank}.json"), "w") as fh:
        json.dump(stats, fh)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
