#!/usr/bin/env python3
"""Drop-in for the reference's trainscripts/uce_sd_debias.py (same flags, prints and artifact)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from uce_amd import cli, debias  # noqa: E402
from uce_amd.sd import pipeline as sdp  # noqa: E402

torch.set_grad_enabled(False)


def main(argv=None) -> None:
    args = cli.parse_debias_args(argv)
    job = cli.debias_job_from_args(args)
    os.makedirs(job.save_dir, exist_ok=True)
    for line in job.banner:
        print(line)
    # one process per GPU under torch.distributed.run: the pipeline, the classifier and the solve all live on THIS
    # rank's device (resolved before anything is loaded, as generate.generate_images does)
    device = debias.rank_device(job.device)
    pipe = sdp.load_pipeline(job.model_id, torch_dtype=torch.float32, device=device, model_dir=args.model_dir,
                             synthetic=args.synthetic_model, vae=True)
    classify = debias.clip_zero_shot_classifier(device)
    debias.UCE(pipe, classify, job.edit_concepts, job.debias_concepts, job.preserve_concepts, job.edit_scale,
               job.preserve_scale, job.lamb, job.save_dir, job.exp_name, job.max_diff, job.step_size,
               job.num_images_per_prompt, job.num_inference_steps, job.guidance_scale,
               desired_ratios=job.desired_ratios, max_iterations=job.max_iterations, device=device,
               algo=cli.ALGO_IDS[args.algo], embed_batch=args.embed_batch)


if __name__ == "__main__":
    main()
