"""Element-wise passes of the transformer blocks at the generation batch (N = 256 images): GEGLU and LayerNorm (with and without the
residual join), us per call and TB/s of their algorithmic bytes.  python tools/probe_ew.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
out = {}
for hw, C in ((64, 320), (32, 640), (16, 1280)):
    rows = 256 * hw * hw
    x = torch.randn(rows, 8 * C, device="cuda").to(torch.bfloat16)
    ms = bench.time_kernel(lambda: H.geglu(x), 20)
    out[f"geglu_{hw}x{hw}_C{C}"] = (round(ms * 1e3, 1), round(x.numel() * 2 * 1.5 / ms / 1e9, 2))
    del x
    a = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    r = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    w = torch.randn(C, device="cuda").to(torch.bfloat16)
    ms = bench.time_kernel(lambda: H.layernorm(a, w, w, 1e-5), 20)
    out[f"ln_{hw}x{hw}_C{C}"] = (round(ms * 1e3, 1), round(a.numel() * 2 * 2 / ms / 1e9, 2))
    ms = bench.time_kernel(lambda: H.layernorm(a, w, w, 1e-5, residual=r), 20)
    out[f"ln_res_{hw}x{hw}_C{C}"] = (round(ms * 1e3, 1), round(a.numel() * 2 * 4 / ms / 1e9, 2))
    del a, r
print(json.dumps(out))
