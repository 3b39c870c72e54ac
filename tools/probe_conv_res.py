"""GPU probe: the implicit-GEMM convolution with and without the residual epilogue at the generation batch (us per launch).
Run once per library (UCE_HIP_LIB) for a same-box A/B of an epilogue change.  Usage: python tools/probe_conv_res.py [CFG batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402
from probe_r04 import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = E.UceHandle("cuda:0")
for Cin, Cout, hw in ((320, 320, 64), (640, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8)):
    x = torch.randn(B, Cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda").bfloat16()
    r = torch.randn(B, Cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    ent = {"lib": os.path.basename(os.environ.get("UCE_HIP_LIB", "default")), "N": B, "Cin": Cin, "Cout": Cout, "H": hw}
    ent["plain_us"] = round(timeit(lambda: H.conv3x3_igemm(x, w, b)), 1)
    ent["bias_none_us"] = round(timeit(lambda: H.conv3x3_igemm(x, w, None)), 1)
    ent["res_us"] = round(timeit(lambda: H.conv3x3_igemm(x, w, b, residual=r)), 1)
    ent["plain2_us"] = round(timeit(lambda: H.conv3x3_igemm(x, w, b)), 1)
    print(json.dumps(ent), flush=True)
