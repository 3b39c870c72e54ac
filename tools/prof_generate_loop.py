"""GPU probe for rocprofv3: kernels of the denoising loop at B prompts per U-Net call (eager launches,
so each one is attributed).  Usage: prof_generate_loop.py [steps] [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd.sd import pipeline as sdp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
pipe.use_graph = False
prompts = [f"a photo {i}" for i in range(B)]
gens = lambda: [torch.Generator().manual_seed(i) for i in range(B)]
pipe(prompts, num_inference_steps=1, generator=gens())
torch.cuda.synchronize(); t0 = time.time()
pipe(prompts, num_inference_steps=steps, generator=gens())
torch.cuda.synchronize(); print("eager", time.time() - t0)
