"""GPU probe: MIOpen exhaustive find (cudnn.benchmark) on/off at the batched U-Net call, prompts per call 8 / 16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd.sd import pipeline as sdp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)

def run(B, out="latent"):
    prompts = [f"a photo {i}" for i in range(B)]
    gens = [torch.Generator().manual_seed(i) for i in range(B)]
    return pipe(prompts, num_inference_steps=steps, output_type=out, generator=gens)

for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    pipe._graphs.clear()
    for B in (8, 16):
        t0 = time.time(); run(B); torch.cuda.synchronize(); first = time.time() - t0
        t0 = time.time(); run(B); torch.cuda.synchronize(); dt = time.time() - t0
        run(B, "pil")
        t0 = time.time(); run(B, "pil"); torch.cuda.synchronize(); full = time.time() - t0
        print(f"find={'exhaustive' if bench else 'default'} B={B:2d}: first {first:6.1f} s, U-Net loop {dt*1e3:7.1f} ms -> {B/dt:5.2f} images/s; "
              f"with VAE + PIL {full*1e3:7.1f} ms -> {B/full:5.2f} images/s", flush=True)
