"""k_conv3x3_w1 persistent (one workgroup per CU, the next tile's first k-tile under the epilogue: UCE_CONV_W1=1, by rule) against one
workgroup per tile (UCE_CONV_W1=5) on the U-Net's 3 x 3 convolutions at the generation batch: us per launch, equal bits.
   python tools/probe_conv_persist.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from uce_amd import edit as E  # noqa: E402


def handle(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return E.UceHandle("cuda:0")
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Hp, H1 = handle(UCE_CONV_W1=1), handle(UCE_CONV_W1=5)
g = torch.Generator(device="cuda").manual_seed(3)
for N, Cin, Cout, Hh, Ww, res in ((B, 320, 320, 64, 64, True), (B, 640, 320, 64, 64, False), (B, 960, 320, 64, 64, False),
                                  (B, 640, 640, 32, 32, True), (B, 1280, 640, 32, 32, False), (B, 1920, 640, 32, 32, False),
                                  (B, 1280, 1280, 16, 16, True), (B, 2560, 1280, 16, 16, False), (16, 512, 512, 256, 256, False)):
    x = torch.randn(N, Cin, Hh, Ww, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (9 * Cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda", generator=g).bfloat16()
    r = torch.randn(N, Cout, Hh, Ww, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    yp, y1 = Hp.conv3x3_igemm(x, w, b, residual=r), H1.conv3x3_igemm(x, w, b, residual=r)
    tp = bench.time_kernel(lambda: Hp.conv3x3_igemm(x, w, b, residual=r), 6) * 1e3
    t1 = bench.time_kernel(lambda: H1.conv3x3_igemm(x, w, b, residual=r), 6) * 1e3
    tp2 = bench.time_kernel(lambda: Hp.conv3x3_igemm(x, w, b, residual=r), 6) * 1e3
    fl = 2.0 * N * Hh * Ww * 9 * Cin * Cout
    print(json.dumps({"N": N, "Cin": Cin, "Cout": Cout, "HW": Hh, "res": res, "persist_us": round(tp, 1), "tile_us": round(t1, 1),
                      "persist2_us": round(tp2, 1), "persist_PFs": round(fl / tp / 1e9, 3), "tile_PFs": round(fl / t1 / 1e9, 3),
                      "equal": bool(torch.equal(yp, y1))}), flush=True)
    del x, w, b, r, yp, y1
