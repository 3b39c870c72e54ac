"""Worker for test_debias_sampling_is_sharded_over_ranks: runs uce_amd.debias.get_ratios on the tiny
CPU pipeline under torch.distributed (gloo) and saves this rank's view of the direction_scale matrix."""
import os
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import debias, generate  # noqa: E402
from uce_amd import edit as E  # noqa: E402
from uce_amd.sd import pipeline as sdp  # noqa: E402

out_dir = sys.argv[1]
rank, world, _ = generate.dist_env()
generate.init_distributed(torch.device("cpu"))
pipe = sdp.load_pipeline("tiny-sd-test", torch.float32, "cpu", synthetic=True, vae=True)
calls = []


class SeededPipe:
    """The reference samples unseeded; to compare ranks the test seeds the global RNG per concept."""
    unet = pipe.unet

    def __call__(self, concept, **kw):
        calls.append(concept)
        torch.manual_seed(zlib.crc32(concept.encode()))
        return pipe(concept, **kw)


def classify(images, labels):
    return [labels[int(np.asarray(im)[..., 0].mean() > np.asarray(im)[..., 1].mean())] for im in images]


slab = E.WeightSlab.from_modules(E.collect_uce_modules(pipe.unet), "cpu")
edit = ["doctor", "nurse", "teacher", "pilot", "chef"]
ds = debias.get_ratios(SeededPipe(), classify, slab, edit, ["male", "female"], [0.5, 0.5], 0.0,
                       num_images_per_prompt=4, num_inference_steps=2, rank=rank, world=world)
np.save(os.path.join(out_dir, f"ratios_w{world}_r{rank}.npy"), ds)
with open(os.path.join(out_dir, f"calls_w{world}_r{rank}.txt"), "w") as f:
    f.write(";".join(calls))
