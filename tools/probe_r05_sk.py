"""Few-tile forms per layer shape of the SD-1.4 U-Net at ONE prompt per call (CFG batch 2), on the GPU box: the split-contraction forms
of uce_linear_fwd / uce_conv3x3_nhwc_fwd (ring of 2 / 3 / 4 stages, 128 x 64 / 128 x 128 tiles) against the GEMM library (torch) and
im2col + library GEMM.  Every timing is a hipGraph of 20 back-to-back launches (a Python launch costs more than these kernels run).
Usage: python tools/probe_r05_sk.py [gemm] [conv] [B=1]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402


def timeit_graph(fn, reps=20, iters=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


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


def gemm(B):
    shapes = []
    for hw, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        M = 2 * B * hw
        shapes += [(M, C, C, "proj", 0), (M, 3 * C, C, "qkv", 0), (M, 8 * C, C, "ff_proj", 1), (M, C, 4 * C, "ff_out", 0)]
    shapes += [(2 * B, 1280, 320, "time1", 0), (2 * B, 1280, 1280, "time2", 0), (2 * B * 77, 1280, 768, "ctx_kv", 0)]
    forms = (0, 7128064, 8128064, 9128064, 7128128, 8128128, 9128128, 128320, 64128320)
    Hs = {f: handle(UCE_GEMM_TILE=f) for f in forms}
    for M, N, K, tag, geglu in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        ent = {"M": M, "N": N, "K": K, "tag": tag, "gflop": round(2e-9 * M * N * K, 2)}
        if geglu:
            ent["torch_us"] = timeit_graph(lambda: Hs[0].geglu(F.linear(x, w, b)))
        else:
            ent["torch_us"] = timeit_graph(lambda: F.linear(x, w, b))
        for f in forms:
            try:
                ent[f"t{f}_us"] = timeit_graph(lambda: Hs[f].linear(x, w, b, geglu=bool(geglu)))
            except Exception as err:  # noqa: BLE001
                ent[f"t{f}_us"] = None
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in ent.items()}), flush=True)


def conv(B):
    N = 2 * B
    cases = [(N, 320, 320, 64, 64, 1, 0), (N, 640, 640, 32, 32, 1, 0), (N, 1280, 1280, 16, 16, 1, 0), (N, 2560, 1280, 16, 16, 1, 0),
             (N, 1280, 1280, 8, 8, 1, 0), (N, 2560, 1280, 8, 8, 1, 0), (N, 1920, 640, 32, 32, 1, 0), (N, 960, 320, 64, 64, 1, 0),
             (N, 320, 320, 64, 64, 2, 0), (N, 640, 640, 32, 32, 2, 0), (N, 1280, 1280, 16, 16, 2, 0),
             (N, 640, 320, 64, 64, 1, 0), (N, 320, 640, 32, 32, 1, 0), (N, 1280, 640, 32, 32, 1, 0), (N, 640, 1280, 16, 16, 1, 0),
             (N, 1280, 1280, 16, 16, 1, 1), (N, 640, 640, 32, 32, 1, 1)]
    forms = (0, 7128064, 8128064, 9128064, 7128128, 8128128, 9128128, 128320, 64128320)
    Hs = {f: handle(UCE_CONV_TILE=f) for f in forms}
    for Nn, Cin, Cout, Hh, Ww, stride, up in cases:
        x = torch.randn(Nn, Cin, Hh, Ww, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda").bfloat16()
        Ho, Wo = (2 * Hh, 2 * Ww) if up else (Hh // stride, Ww // stride)
        ent = {"N": Nn, "Cin": Cin, "Cout": Cout, "H": Hh, "W": Ww, "stride": stride, "up": up,
               "gflop": round(2e-9 * Nn * Ho * Wo * 9 * Cin * Cout, 2)}
        if stride == 1 and not up:
            wmat = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
            cols = torch.empty(Nn * Hh * Ww, 9 * Cin, device="cuda", dtype=torch.bfloat16)
            y = torch.empty(Nn * Hh * Ww, Cout, device="cuda", dtype=torch.bfloat16)
            H0 = Hs[0]

            def im2col_lib():
                H0.lib.uce_im2col3x3_nhwc(H0._h, x.data_ptr(), cols.data_ptr(), Nn, Hh, Ww, Cin, 0, torch.cuda.current_stream().cuda_stream)
                torch.addmm(b, cols, wmat.t(), out=y)
            ent["im2col_lib_us"] = timeit_graph(im2col_lib)
        for f in forms:
            try:
                ent[f"t{f}_us"] = timeit_graph(lambda: Hs[f].conv3x3_igemm(x, w, b, stride=stride, upsample=bool(up)))
            except Exception as err:  # noqa: BLE001
                ent[f"t{f}_us"] = None
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in ent.items()}), flush=True)


def sweep(B):
    """slabs per tile: UCE_SK_SPLIT = 1 .. 16 on both few-tile forms"""
    splits = (1, 2, 3, 4, 6, 8, 12, 16)
    gs = [(4096 * 2 * B, 320, 1280), (1024 * 2 * B, 640, 640), (1024 * 2 * B, 640, 2560), (256 * 2 * B, 1280, 1280), (256 * 2 * B, 1280, 5120),
          (64 * 2 * B, 1280, 1280), (64 * 2 * B, 1280, 5120), (64 * 2 * B, 3840, 1280), (2 * B, 1280, 1280)]
    cs = [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8), (2560, 1280, 16), (2560, 1280, 8), (960, 320, 64)]
    for form in [int(f) for f in os.environ.get('UCE_PROBE_FORMS', '7128064,7128128').split(',')]:
        Hs = {S: handle(UCE_GEMM_TILE=form, UCE_CONV_TILE=form, UCE_SK_SPLIT=S) for S in splits}
        for M, N, K in gs:
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            ent = {"form": form, "gemm": [M, N, K]}
            for S in splits:
                if S <= K // 64:
                    ent[f"S{S}"] = round(timeit_graph(lambda: Hs[S].linear(x, w, b)), 1)
            print(json.dumps(ent), flush=True)
        for Cin, Cout, Hh in cs:
            x = torch.randn(2 * B, Cin, Hh, Hh, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
            w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
            b = torch.randn(Cout, device="cuda").bfloat16()
            ent = {"form": form, "conv": [Cin, Cout, Hh]}
            for S in splits:
                ent[f"S{S}"] = round(timeit_graph(lambda: Hs[S].conv3x3_igemm(x, w, b)), 1)
            print(json.dumps(ent), flush=True)
        for h_ in Hs.values():
            h_.close()


def sattn(B):
    """self-attention forms (UCE_SATTN_QT x UCE_SATTN_VTI) at the four attn1 shapes of one prompt per call"""
    for L, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        qkv = torch.randn(2 * B, L, 3 * C, device="cuda").bfloat16()
        ent = {"sattn": [2 * B, L, C], "gflop": round(4e-9 * 2 * B * L * L * C, 2)}
        for qt in (0, 1, 2, 3, 4):
            for vti in (0, 1, 2, 3):
                Hv = handle(UCE_SATTN_QT=qt, UCE_SATTN_VTI=vti)
                try:
                    ent[f"qt{qt}_vti{vti}"] = round(timeit_graph(lambda: Hv.sattn_packed(qkv, 8)), 1)
                except Exception as err:  # noqa: BLE001
                    ent[f"qt{qt}_vti{vti}"] = None
                torch.cuda.synchronize()
                Hv.close()
        print(json.dumps(ent), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.isdigit()] or ["gemm", "conv"]
    Bp = [int(a) for a in sys.argv[1:] if a.isdigit()]
    B = Bp[0] if Bp else 1
    if "gemm" in args:
        gemm(B)
    if "conv" in args:
        conv(B)
    if "sweep" in args:
        sweep(B)
    if "sattn" in args:
        sattn(B)
