"""GPU probe: the four-wave 128 x 160 form (UCE_GEMM_TILE=4128160, two workgroups per CU) against the rule's form on the K = 320 / 640
layers at CFG batch 256 (us per launch; bias, bias + residual), with a correctness check against the rule's output."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probe_r04 import handle, timeit  # noqa: E402

H0 = handle(UCE_GEMM_TILE=0)
H4 = handle(UCE_GEMM_TILE=4128160)
for M, N, K in ((1048576, 320, 320), (1048576, 960, 320), (1048576, 320, 1280), (262144, 640, 640), (262144, 1920, 640), (1000, 328, 320)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16()
    y0, y4 = H0.linear(x, w, b, r), H4.linear(x, w, b, r)
    ent = {"M": M, "N": N, "K": K, "max_abs_diff": float((y0.float() - y4.float()).abs().max()), "equal": bool(torch.equal(y0, y4))}
    for name, Hh in (("rule", H0), ("w4", H4), ("rule2", H0), ("w4_2", H4)):
        ent[name + "_us"] = round(timeit(lambda: Hh.linear(x, w, b)), 1)
        ent[name + "_res_us"] = round(timeit(lambda: Hh.linear(x, w, b, r)), 1)
    print(json.dumps(ent), flush=True)
