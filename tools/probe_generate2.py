"""GPU probe: images/s of the synthetic SD-1.4 pipeline vs prompts per U-Net call, hipGraph on/off;
start-up (first call: MIOpen solver search) and VAE decode + PNG times.  Usage: probe_generate2.py [steps] [batches]"""
import sys, os, time, traceback, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd.sd import pipeline as sdp

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16]
dev = "cuda:0"
t0 = time.time()
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, dev, synthetic=True, vae=True)
print(f"build {time.time() - t0:.1f} s", flush=True)


def run(B, graph, out="latent"):
    pipe.use_graph = graph
    prompts = [f"a photo {i}" for i in range(B)]
    gens = [torch.Generator().manual_seed(i) for i in range(B)]
    return pipe(prompts if B > 1 else prompts[0], num_inference_steps=steps, output_type=out,
                generator=gens if B > 1 else gens[0])

for B in batches:
    for graph in (False, True):
        try:
            t0 = time.time(); run(B, graph); torch.cuda.synchronize(); first = time.time() - t0
            t0 = time.time(); run(B, graph); torch.cuda.synchronize(); dt = time.time() - t0
            print(f"B={B:3d} graph={graph!s:5s}: first call {first:6.1f} s, steady {dt*1e3:8.1f} ms -> {B/dt:6.2f} images/s "
                  f"({dt/ (steps+1) *1e3:6.2f} ms per U-Net call)", flush=True)
        except Exception:
            print(f"B={B} graph={graph}: FAILED"); traceback.print_exc()
    try:
        t0 = time.time(); o = run(B, True, "pil"); torch.cuda.synchronize(); dt = time.time() - t0
        t1 = time.time()
        for im in o.images:
            im.save(io.BytesIO(), format="PNG")
        print(f"B={B:3d} full (graph + VAE + PIL): {dt*1e3:8.1f} ms -> {B/dt:6.2f} images/s ; PNG encode {1e3*(time.time()-t1)/B:.1f} ms/image", flush=True)
    except Exception:
        traceback.print_exc()
print("max memory GB", torch.cuda.max_memory_allocated() / 2**30)
