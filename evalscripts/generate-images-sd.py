#!/usr/bin/env python3
"""Drop-in for the reference's evalscripts/generate-images-sd.py (same flags and output layout).
Single process: `python evalscripts/generate-images-sd.py --prompts_path ...`.
One process per GPU:  `python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1
evalscripts/generate-images-sd.py ...` - rank 0 broadcasts the edited weights (RCCL over xGMI), the
CSV rows are then sharded round-robin over the ranks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from uce_amd import cli, generate  # noqa: E402

torch.set_grad_enabled(False)


def main(argv=None) -> None:
    a = cli.parse_generate_args(argv)
    stats = generate.generate_images(
        model_id=a.model_id, uce_model_path=a.uce_model_path, prompts_path=a.prompts_path, save_path=a.save_path,
        exp_name=a.exp_name, device=a.device, torch_dtype=torch.bfloat16, guidance_scale=a.guidance_scale,
        num_inference_steps=a.num_inference_steps, num_images_per_prompt=a.num_images_per_prompt,
        from_case=a.from_case, till_case=a.till_case, model_dir=a.model_dir, synthetic=a.synthetic_model,
        latents_only=a.latents_only, skip_existing=a.skip_existing, batch_prompts=a.batch_prompts)
    if int(stats["rank"]) == 0 and stats["seconds"] > 0:
        total = stats.get("images_total", stats["images"])
        secs = stats.get("seconds_max", stats["seconds"])
        print(f"generated {int(total)} images in {secs:.2f} s on {int(stats['world'])} GPU(s): {total / max(secs, 1e-9):.3f} images/s")


if __name__ == "__main__":
    main()
