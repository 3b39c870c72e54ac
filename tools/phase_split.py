"""Where an image's 106 ms go at the bench's batch: the same 128-prompt call with and without the VAE decode / image conversion, and
with 1 instead of 50 steps (text encoder + fixed costs).  Usage (GPU box): python tools/phase_split.py [B=128]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd.sd import pipeline as sdp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
prompts = [f"a photo of thing number {i}" for i in range(B)]
gens = lambda: [torch.Generator().manual_seed(i) for i in range(B)]


def run(steps, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe(prompts, num_inference_steps=steps, guidance_scale=7.5, generator=gens(), **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run(2)
run(2, output_type="latent")
res = {"B": B}
res["full_50_s"] = min(run(50), run(50))
res["latent_50_s"] = min(run(50, output_type="latent"), run(50, output_type="latent"))
res["full_2_s"] = min(run(2), run(2))
res["latent_2_s"] = min(run(2, output_type="latent"), run(2, output_type="latent"))
res["per_image_ms"] = {"total": 1e3 * res["full_50_s"] / B, "vae_and_conversion": 1e3 * (res["full_50_s"] - res["latent_50_s"]) / B,
                       "per_unet_call": 1e3 * (res["latent_50_s"] - res["latent_2_s"]) / 48 / B,
                       "fixed_text_encoder_latents": 1e3 * (res["latent_2_s"] - 3 * (res["latent_50_s"] - res["latent_2_s"]) / 48) / B}
print(json.dumps(res))
