"""uce_sattn_packed_fwd at the small batches of one to eight prompts per call (CFG batch 2 .. 16), SD-1.4's three streaming shapes: us
per launch.  Run under UCE_SATTN_QT = 0 (rule) / 1 / 2 / 4 to compare the kernel forms:  python tools/probe_sattn_small_batch.py"""
import json, sys, torch
sys.path.insert(0, ".")
import bench
from uce_amd import edit as E
H = E.UceHandle.get("cuda:0")
out = {}
for B in (2, 4, 8, 16):
    for L, dh in ((4096, 40), (1024, 80), (256, 160)):
        C = 8 * dh
        qkv = torch.randn(B, L, 3 * C, device="cuda").bfloat16()
        ms = bench.time_kernel(lambda: H.sattn_packed(qkv, 8), 20)
        out[f"B{B}_L{L}"] = round(ms * 1e3, 1)
print(json.dumps(out))
