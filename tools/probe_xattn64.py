"""Cross-attention at SDXL's head dim (dh = 64) at the generation batch: the column-group kernel (default) against the
per-(batch, head) kernel (UCE_XATTN_VARIANT=0), us per launch and fraction of 8 TB/s on the Q + O + K + V bytes (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def handle(variant):
    from uce_amd import edit as E
    os.environ["UCE_XATTN_VARIANT"] = variant
    try:
        return E.UceHandle("cuda:0")
    finally:
        del os.environ["UCE_XATTN_VARIANT"]


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


Hs = {"group (default)": handle("1"), "per (b, h)": handle("0")}
for B in (32, 128):
    for (Lq, C) in ((4096, 640), (1024, 1280)):
        H_ = C // 64
        q = torch.randn(B, Lq, C, device="cuda").bfloat16()
        k = torch.randn(B, 77, C, device="cuda").bfloat16()
        v = torch.randn_like(k)
        o = torch.empty_like(q)
        nbytes = 2 * (2 * q.numel() + 2 * k.numel())
        row = {n: timed(lambda: h.xattn(q, k, v, H_, out=o)) for n, h in Hs.items()}
        print(f"B={B} Lq={Lq} C={C} dh=64: " + " | ".join(f"{n} {us:.1f} us = {nbytes / us / 8e6:.2f} of 8 TB/s" for n, us in row.items()))
