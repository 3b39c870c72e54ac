"""GPU probe: uce_sattn_fwd vs torch SDPA at the SD-1.4 attn1 shapes (time, effective TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (2, 32):
    for L, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
        C = 8 * dh
        q = torch.randn(B, L, C, device="cuda").bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q)
        o = torch.empty_like(q)
        sp = lambda t: t.view(B, L, 8, dh).transpose(1, 2)
        t_hip = timeit(lambda: H.sattn(q, k, v, 8, out=o))
        t_sdpa = timeit(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, L, C))
        fl = 4.0 * B * 8 * L * L * dh
        print(f"B={B:2d} L={L:4d} dh={dh:3d}: hip {t_hip:9.1f} us ({fl/t_hip/1e6:7.1f} TF/s)   torch SDPA {t_sdpa:9.1f} us ({fl/t_sdpa/1e6:7.1f} TF/s)", flush=True)
