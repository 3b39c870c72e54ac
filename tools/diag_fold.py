"""Diagnostic of k_sattn_h<FOLD> (prescaled q) over score magnitudes and key counts: non-finite outputs, error against fp64,
where the first bad rows sit (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

os.environ["UCE_SATTN_QT"] = "4"
from uce_amd import edit as E

LOG2E = 1.4426950408889634
H = E.UceHandle("cuda:0")
dh, H_ = 40, 4
C = H_ * dh
c = dh ** -0.5 * LOG2E
shown = False
for L in (32, 64, 128, 640):
    for mag in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0):
        g = torch.Generator().manual_seed(L * 3 + int(mag * 10))
        qp = (torch.randn(1, L, C, generator=g) * mag * c).to(torch.bfloat16)
        k = (torch.randn(1, L, C, generator=g) * mag).to(torch.bfloat16)
        v = torch.randn(1, L, C, generator=g).to(torch.bfloat16)
        o = H.sattn_packed(torch.cat([qp, k, v], dim=-1).cuda(), H_, prescaled=True).float().cpu()
        sp = lambda t: t.double().view(1, L, H_, dh).transpose(1, 2)
        s = sp(qp) @ sp(k).transpose(-1, -2) / LOG2E
        ref = (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(1, L, C)
        bad = ~torch.isfinite(o)
        fin = torch.where(bad, torch.zeros_like(o), o)
        err = ((fin.double() - torch.where(bad, torch.zeros_like(ref), ref)).norm() / ref.norm()).item()
        rows = sorted(set(bad.nonzero()[:, 1].tolist()))[:12]
        heads = sorted(set((bad.nonzero()[:, 2] // dh).tolist()))
        smax = (s * LOG2E).abs().max().item()
        print(f"L={L} mag={mag} max|s c|={smax:.1f} nonfinite={int(bad.sum())} relF(finite part)={err:.2e} bad rows {rows} heads {heads}")
        if bad.any() and not shown:
            shown = True
            s2 = s * LOG2E                                   # [1, H, L, L] scores in log2 units
            for (r, hd) in sorted(set(zip(bad.nonzero()[:, 1].tolist(), (bad.nonzero()[:, 2] // dh).tolist())))[:16]:
                row = s2[0, hd, r]
                halves = row.view(-1, 32).max(1).values if L % 32 == 0 else row[:32].max().view(1)
                vals = o[0, r, hd * dh:hd * dh + 4].tolist()
                print(f"   row {r} head {hd}: max {row.max():.1f} at key {int(row.argmax())}, min {row.min():.1f}, half maxima "
                      f"{[round(float(x), 0) for x in halves[:8]]}, out {vals}")
