#!/usr/bin/env python3
"""Drop-in for the reference's trainscripts/uce_hidream_edit.py (same flags, prints and artifact)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from uce_amd import cli, hidream  # noqa: E402

torch.set_grad_enabled(False)


def main(argv=None) -> None:
    args = cli.parse_hidream_args(argv)
    job = cli.erase_job_from_args(args)          # same concept-list semantics as the SD script (:216-270)
    os.makedirs(job.save_dir, exist_ok=True)
    for line in job.banner:
        print(line)
    hidream.UCE(job.model_id, job.edit_concepts, job.guide_concepts, job.preserve_concepts, job.erase_scale,
                job.preserve_scale, job.lamb, job.save_dir, job.exp_name, torch.float32, job.device,
                cli.HIDREAM_MAX_SEQUENCE_LENGTH, algo=cli.ALGO_IDS[args.algo])


if __name__ == "__main__":
    main()
