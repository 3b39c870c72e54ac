"""GPU probe: uce_layernorm_fwd against torch's LayerNorm kernel (and against add + LayerNorm for the fused residual
join) on the transformer-block shapes of the SD-1.4 U-Net at batch N.  Usage: probe_layernorm.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from uce_amd import edit as E

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = E.UceHandle.get("cuda:0")


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


for L, C in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
    x = torch.randn(N, L, C, device="cuda:0", dtype=torch.bfloat16)
    r = torch.randn(N, L, C, device="cuda:0", dtype=torch.bfloat16)
    w = torch.randn(C, device="cuda:0", dtype=torch.bfloat16)
    b = torch.randn(C, device="cuda:0", dtype=torch.bfloat16)
    nbytes = x.numel() * 2
    t_hip = timeit(lambda: H.layernorm(x, w, b, 1e-5))
    t_torch = timeit(lambda: F.layer_norm(x, (C,), w, b, 1e-5))
    t_hip_f = timeit(lambda: H.layernorm(x, w, b, 1e-5, residual=r))
    t_torch_f = timeit(lambda: F.layer_norm(x + r, (C,), w, b, 1e-5))
    print(f"N={N} L={L:4d} C={C:4d}: LN hip {t_hip:6.1f} us ({2 * nbytes / t_hip / 1e6:4.2f} TB/s) torch {t_torch:6.1f} us | "
          f"add+LN hip {t_hip_f:6.1f} us ({4 * nbytes / t_hip_f / 1e6:4.2f} TB/s) torch {t_torch_f:6.1f} us", flush=True)
