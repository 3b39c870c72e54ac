#!/bin/bash
# A/B of the one-prompt-per-call denoising loop under a hipGraph (wall per U-Net step): env assignments given as arguments, one
# configuration per argument ("A=1 B=2"), "-" = defaults.  Usage (GPU box): bash tools/ab_b1.sh "-" "UCE_CONV_TILE=8128064" ...
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from uce_amd.sd import pipeline as sdp
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
g = lambda: torch.Generator().manual_seed(1)
pipe("a photo", num_inference_steps=2, generator=g())
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    pipe("a photo", num_inference_steps=50, generator=g())
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("AB", os.environ.get("AB_LABEL", ""), {k: v for k, v in os.environ.items() if k.startswith("UCE_")}, "image_s %.4f  images_per_s %.3f" % (best, 1.0 / best), flush=True)
PY
done
