"""A few launches of uce_linear_fwd at named shapes, for rocprofv3 --pmc passes (tools/r04_c.sh).
Usage: probe_gemm_pmc.py  (fixed list; UCE_GEMM_TILE in the environment pins a tile)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402

H = E.UceHandle.get("cuda:0")
for M, N, K in ((131072, 2560, 320), (32768, 5120, 640), (8192, 3840, 1280), (8192, 1280, 5120), (131072, 320, 320)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(4):
        H.linear(x, w, b)
    torch.cuda.synchronize()
