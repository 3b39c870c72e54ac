"""GPU probe: which convolutions of the pipeline still reach the library (F.conv2d), with shapes and synchronous timings."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from uce_amd.sd import pipeline as sdp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=True)
pipe.use_graph = False
prompts = [f"a photo {i}" for i in range(B)]
gens = lambda: [torch.Generator().manual_seed(i) for i in range(B)]
pipe(prompts, num_inference_steps=1, generator=gens())
torch.cuda.synchronize()
log = collections.defaultdict(lambda: [0, 0.0])
orig = F.conv2d
def spy(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    y = orig(x, w, b, stride, padding, dilation, groups)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    key = (tuple(x.shape), tuple(w.shape), str(stride), x.is_contiguous(memory_format=torch.channels_last), w.is_contiguous(memory_format=torch.channels_last), str(x.dtype))
    log[key][0] += 1; log[key][1] += dt
    return y
F.conv2d = spy
torch.nn.functional.conv2d = spy
pipe(prompts, num_inference_steps=2, generator=gens())
for k, (n, t) in sorted(log.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:3d} x {t / n * 1e6:9.1f} us  {k}")
