"""Self-attention error against fp64 of the half-tile pipeline (k_sattn_h forced), a few sizes and both element types - printed,
not asserted: for same-box comparisons of kernel forms (UCE_SATTN_REL=0|1 while the A/B switch exists)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UCE_SATTN_QT"] = "4"
from uce_amd import edit as E  # noqa: E402


def ref(q, k, v, heads):
    B, L, C = q.shape
    dh = C // heads
    sp = lambda t: t.double().view(B, -1, heads, dh).transpose(1, 2)  # noqa: E731
    s = sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5
    return (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, -1, C)


H = E.UceHandle("cuda:0")
for (B, H_, Lq, Lk, dh, dt, mul) in [(3, 4, 77, 64, 40, torch.float16, 1), (1, 4, 257, 128, 40, torch.float16, 1),
                                      (2, 8, 1024, 1024, 40, torch.bfloat16, 1), (2, 8, 1024, 1024, 40, torch.float16, 1),
                                      (1, 8, 4096, 4096, 40, torch.bfloat16, 1), (1, 8, 4096, 4096, 40, torch.bfloat16, 5),
                                      (1, 8, 4096, 4096, 40, torch.float16, 3)]:
    g = torch.Generator().manual_seed(Lq * 5 + dh + Lk)
    C = H_ * dh
    q = (mul * torch.randn(B, Lq, C, generator=g)).to(dt).cuda()
    k = (mul * torch.randn(B, Lk, C, generator=g)).to(dt).cuda()
    v = torch.randn(B, Lk, C, generator=g).to(dt).cuda()
    o = H.sattn(q, k, v, H_)
    r = ref(q, k, v, H_)
    sp = lambda t: t.view(B, -1, H_, dh).transpose(1, 2)  # noqa: E731
    ot = torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, -1, C)
    print(B, H_, Lq, Lk, dh, dt, "x%d" % mul, "relF %.3e" % float((o.double() - r).norm() / r.norm()),
          "torch SDPA relF %.3e" % float((ot.double() - r).norm() / r.norm()))
