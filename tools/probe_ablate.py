"""Time uce_linear_fwd on a few large shapes with whatever library UCE_HIP_LIB names (measurement builds of uce_gemm.hip,
-DUCE_GEMM_ABLATE=n) and whatever tile form UCE_GEMM_TILE forces.  SHAPES="MxNxK,..." overrides the list."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E
H = E.UceHandle("cuda:0")
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [(131072, 320, 1280), (131072, 320, 320), (32768, 640, 2560), (131072, 960, 320), (32768, 1920, 640), (524288, 320, 320)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
out = []
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.zeros(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    try:
        us = timeit(lambda: H.linear(x, w, b, out=y))
        out.append(f"{M}x{N}x{K}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF/s")
    except Exception as e:  # noqa: BLE001
        out.append(f"{M}x{N}x{K}: {type(e).__name__}")
print(os.environ.get("TAG", "?"), " | ".join(out), flush=True)
