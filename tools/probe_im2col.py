"""GPU probe: the patch-matrix kernel alone and the whole conv3x3 (im2col + hipBLASLt GEMM) per U-Net conv shape at
U-Net batch N, for each UCE_IM2COL_VARIANT.  Usage: probe_im2col.py [N] [variants, comma separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E, lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
H = E.UceHandle.get("cuda:0")
SHAPES = [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (960, 320, 64), (640, 320, 64), (1920, 640, 32),
          (2560, 1280, 16), (1280, 1280, 8), (2560, 1280, 8)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


for cin, cout, hw in SHAPES:
    x = torch.randn(N, cin, hw, hw, device="cuda:0", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda:0", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) * 0.02
    b = torch.randn(cout, device="cuda:0", dtype=torch.bfloat16)
    cols = torch.empty(N * hw * hw, 9 * cin, device="cuda:0", dtype=torch.bfloat16)
    xs = x.permute(0, 2, 3, 1)
    nbytes = cols.numel() * 2
    line = f"N={N} {cin:4d}->{cout:4d} @{hw:2d}: cols {nbytes / 1e6:7.1f} MB |"
    ref = None
    for v in variants:
        os.environ["UCE_IM2COL_VARIANT"] = str(v)
        f = lambda: L.check(H.lib.uce_im2col3x3_nhwc(H._h, xs.data_ptr(), cols.data_ptr(), N, hw, hw, cin, 0,
                                                     torch.cuda.current_stream().cuda_stream), "im2col")
        t = timeit(f)
        if ref is None:
            ref = cols.clone()
        ok = torch.equal(ref, cols)
        tc = timeit(lambda: H.conv3x3_nhwc(x, w, b), 10)
        line += f" v{v}: {t:7.1f} us {nbytes / t / 1e6:5.2f} TB/s conv {tc:7.1f} us{'' if ok else ' MISMATCH'} |"
    print(line, flush=True)
