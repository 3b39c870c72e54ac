"""k_gemm_persist against k_gemm_dma (UCE_GEMM_PERSIST=0) on the linear shapes of the U-Net at the generation batch: us per launch.
   python tools/probe_gemm_persist.py [B]      (run twice: UCE_GEMM_PERSIST=1 / 0)"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from uce_amd import edit as E  # noqa: E402
from uce_amd.sd.unet import geglu_interleave  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H = E.UceHandle.get(dev)
g = torch.Generator(device="cuda").manual_seed(1)
out = {"B": B, "persist": os.environ.get("UCE_GEMM_PERSIST", "1"), "shapes": []}
# (rows per sample, K, N, residual, geglu)
for hw, K, N, res, geglu in ((4096, 320, 320, True, False), (4096, 320, 960, False, False), (4096, 320, 2560, False, True),
                             (4096, 1280, 320, True, False), (1024, 640, 640, True, False), (1024, 640, 1920, False, False),
                             (1024, 640, 5120, False, True), (1024, 2560, 640, True, False), (256, 1280, 1280, True, False),
                             (256, 1280, 10240, False, True), (256, 5120, 1280, True, False)):
    M = B * hw
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g).bfloat16()
    r = torch.randn(M, N, device=dev, generator=g).bfloat16() if res else None
    if geglu:
        w, b = geglu_interleave(w, b)
        fn = lambda: H.linear(x, w, b, geglu=True)  # noqa: E731
        byts = 2.0 * (M * K + N * K + M * N / 2)
    else:
        fn = lambda: H.linear(x, w, b, r)  # noqa: E731
        byts = 2.0 * (M * K + N * K + M * N * (2 if res else 1))
    ms = bench.time_kernel(fn, 10)
    out["shapes"].append({"M": M, "K": K, "N": N, "res": res, "geglu": geglu, "us": round(ms * 1e3, 1),
                          "TBs": round(byts / (ms * 1e-3) / 1e12, 2), "PFs": round(2.0 * M * N * K / (ms * 1e-3) / 1e15, 3)})
    del x, w, b, r
print(json.dumps(out))
