"""GPU probe: images/s of the synthetic SD-1.4 pipeline (bf16, 512x512, PNDM) + xattn kernel timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd.sd import pipeline as sdp
from uce_amd import edit as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H = E.UceHandle.get("cuda:0")
# cross-attention kernel alone at the four SD-1.4 shapes (B = 2: CFG pair)
for Lq, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
    C = 8 * dh
    q = torch.randn(2, Lq, C, device="cuda").bfloat16(); k = torch.randn(2, 77, C, device="cuda").bfloat16(); v = torch.randn_like(k)
    o = torch.empty_like(q)
    for _ in range(3): H.xattn(q, k, v, 8, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): H.xattn(q, k, v, 8, out=o)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    byts = 2 * (2 * Lq * C * 2) + 2 * (2 * 77 * C * 2)
    print(f"xattn Lq={Lq} dh={dh}: {us:7.2f} us  {byts / us / 1e3:8.1f} GB/s  (Q+O+K+V = {byts/1e6:.2f} MB)")
t0 = time.time()
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
print("build", time.time() - t0)
for hoist in (True, False):
    pipe.hoist_context = hoist
    out = pipe("warm up", num_inference_steps=2, generator=torch.Generator().manual_seed(0))
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(nimg):
        out = pipe(f"a photo of an astronaut riding a horse {i}", num_inference_steps=steps, generator=torch.Generator().manual_seed(i))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"hoist_context={hoist}: {nimg} images, {steps} steps: {dt:.2f} s -> {nimg/dt:.3f} images/s")
