"""GPU stress of uce_conv3x3_nhwc_fwd (both implicit-GEMM kernels): random shapes - ragged pixel counts, tiles straddling
images, every output-tile width, fused upsample - against F.conv2d in fp32, each shape launched several times back to back
(the direct-to-LDS ring must never read a stage before its loads have landed).
    python tools/stress_conv.py [shapes] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from uce_amd import edit as E  # noqa: E402

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
H = E.UceHandle.get("cuda:0")
worst = 0.0
for it in range(n_shapes):
    Cin = int(rng.choice([64, 128, 192, 320, 640]))
    Cout = int(rng.choice([64, 128, 256, 320, 384, 512, 640]))
    up = bool(rng.integers(0, 2))
    N = int(rng.integers(1, 4))
    Hh, Ww = int(rng.integers(40, 150)), int(rng.integers(40, 150))
    if up:
        Hh, Ww = Hh & ~1, Ww & ~1
    dtype = torch.float16 if rng.integers(0, 4) == 0 else torch.bfloat16
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    hs, ws = (Hh // 2, Ww // 2) if up else (Hh, Ww)
    x = torch.randn(N, Cin, hs, ws, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", dtype).to(memory_format=torch.channels_last)
    with torch.no_grad():
        xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
        ref = F.conv2d(xin, conv.weight.float(), conv.bias.float(), padding=1)
        outs = [H.conv3x3_igemm(x, conv.weight, conv.bias, upsample=up) for _ in range(4)]
        torch.cuda.synchronize()
    for y in outs:
        err = ((y.float() - ref).norm() / ref.norm()).item()
        worst = max(worst, err)
        if not err < (6e-3 if dtype == torch.bfloat16 else 1e-3) or not torch.equal(y, outs[0]):
            print(f"FAIL shape {it}: N={N} Cin={Cin} Cout={Cout} {Hh}x{Ww} up={up} {dtype}: relF {err:.3e}, "
                  f"repeatable {torch.equal(y, outs[0])}")
            sys.exit(1)
print(f"{n_shapes} shapes x 4 launches, worst relF vs fp32 {worst:.2e}, every repeat bit-identical")
