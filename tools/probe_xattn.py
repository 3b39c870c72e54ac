"""GPU probe: k_xattn at the SD-1.4 shapes, B = 2 and B = 16, time + parity vs torch SDPA (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
for B in (2, 16, 32):
    for Lq, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
        C = 8 * dh
        q = torch.randn(B, Lq, C, device="cuda").bfloat16(); k = torch.randn(B, 77, C, device="cuda").bfloat16(); v = torch.randn_like(k)
        o = torch.empty_like(q)
        H.xattn(q, k, v, 8, out=o)
        sp = lambda t: t.float().view(B, t.shape[1], 8, dh).transpose(1, 2)
        ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Lq, C)
        err = ((o.float() - ref).norm() / ref.norm()).item()
        for _ in range(3): H.xattn(q, k, v, 8, out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): H.xattn(q, k, v, 8, out=o)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        byts = 2 * (B * Lq * C * 2) + 2 * (B * 77 * C * 2)
        print(f"B={B:2d} Lq={Lq:4d} dh={dh:3d}: {us:7.2f} us {byts / us / 1e3:7.1f} GB/s  rel err {err:.2e}", flush=True)
