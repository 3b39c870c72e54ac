#!/usr/bin/env python3
"""Drop-in for the reference's trainscripts/uce_sd_erase.py (same flags, prints and artifact):
erase / moderate concepts in a Stable Diffusion model with the closed-form UCE edit, computed by
hand-written HIP kernels on an MI355X (uce_amd.edit.UCE -> libuce_hip.so)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from uce_amd import cli, edit  # noqa: E402
from uce_amd.sd import pipeline as sdp  # noqa: E402

torch.set_grad_enabled(False)


def main(argv=None) -> None:
    args = cli.parse_erase_args(argv)
    job = cli.erase_job_from_args(args)
    os.makedirs(job.save_dir, exist_ok=True)
    for line in job.banner:
        print(line)
    pipe = sdp.load_pipeline(job.model_id, torch_dtype=torch.float32, device=job.device, model_dir=args.model_dir,
                             synthetic=args.synthetic_model, vae=False)
    edit.UCE(pipe, job.edit_concepts, job.guide_concepts, job.preserve_concepts, job.erase_scale,
             job.preserve_scale, job.lamb, job.save_dir, job.exp_name, device=job.device,
             algo=cli.ALGO_IDS[args.algo], embed_batch=args.embed_batch)


if __name__ == "__main__":
    main()
