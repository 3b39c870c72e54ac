"""GPU probe for rocprofv3: kernels of the VAE decode of B latents (the pipeline's own decode path).  Usage: prof_vae_loop.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd.sd import pipeline as sdp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
lat = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.bfloat16)
img = sdp.images_from_decoded(pipe.vae.decode(lat), "pil")
torch.cuda.synchronize()
t0 = time.time()
img = sdp.images_from_decoded(pipe.vae.decode(lat), "pil")
torch.cuda.synchronize()
print("vae decode of", B, "latents:", round((time.time() - t0) * 1e3, 2), "ms =", round((time.time() - t0) * 1e3 / B, 3), "ms per image", flush=True)
