"""One 3x3 convolution shape through the implicit-GEMM kernel, a few launches (for rocprofv3 --pmc passes):
    python tools/probe_one_conv.py N Cin Cout H [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uce_amd import edit as E  # noqa: E402

N, Cin, Cout, Hh = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
H = E.UceHandle.get("cuda:0")
x = torch.randn(N, Cin, Hh, Hh, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", torch.bfloat16).to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(n):
        H.conv3x3_igemm(x, conv.weight, conv.bias)
torch.cuda.synchronize()
print("done")
