"""Layer census of ONE U-Net call (GPU box): every launch-issuing UceHandle method the SD-1.4 U-Net calls at B prompts per call is
recorded with its shapes, then each distinct call is re-timed alone with HIP events (same shapes / strides, random data).  Prints one
JSON line per distinct call (count per U-Net call, us, GFLOP, TF/s) sorted by its share of the step, and a summary line.
Usage: python tools/probe_r05.py [B=1] [top=40]      (B prompts per call -> CFG batch 2B)"""
import json
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402
from uce_amd.sd import pipeline as sdp  # noqa: E402

METHODS = ["linear", "conv3x3_nhwc", "conv3x3_igemm", "conv3x3_c4", "groupnorm_nhwc", "layernorm", "sattn_packed", "sattn", "xattn",
           "add_bias_nhwc", "geglu", "cfg_pndm_step", "linear_f32", "softmax_rows"]


def sig_of(v):
    if isinstance(v, torch.Tensor):
        return ("T", tuple(v.shape), tuple(v.stride()), str(v.dtype))
    if isinstance(v, (list, tuple)):
        return ("L",) + tuple(sig_of(x) for x in v)
    return ("V", v)


def rebuild(s):
    if s[0] == "T":
        _, shape, stride, dt = s
        t = torch.empty_strided(shape, stride, dtype=getattr(torch, dt.split(".")[1]), device="cuda")
        if t.is_floating_point():
            t.normal_()
        return t
    if s[0] == "L":
        return [rebuild(x) for x in s[1:]]
    return s[1]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def flops(name, args, kw):
    def shp(i):
        return args[i][1] if i < len(args) and args[i][0] == "T" else None
    if name == "linear":
        x, w = shp(0), shp(1)
        M = 1
        for d in x[:-1]:
            M *= d
        K = w[1]
        return 2.0 * M * w[0] * K
    if name in ("conv3x3_nhwc", "conv3x3_igemm"):
        x, w = shp(0), shp(1)
        up = kw.get("upsample", ("V", False))[1]
        st = kw.get("stride", ("V", 1))[1]
        Hh, Ww = (2 * x[2], 2 * x[3]) if up else (x[2] // st, x[3] // st)
        return 2.0 * x[0] * Hh * Ww * 9 * x[1] * w[0]
    if name == "sattn_packed":
        q = shp(0)
        B, L, C3 = q
        return 4.0 * B * L * L * (C3 // 3)
    if name == "xattn":
        q, k = shp(0), shp(1)
        return 4.0 * q[0] * q[1] * k[1] * q[2]
    return 0.0


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, "cuda:0", synthetic=True, vae=False)
    unet = pipe.unet
    x = torch.randn(2 * B, 4, 64, 64, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    t = torch.tensor([500.0] * (2 * B), device="cuda")
    ctx = torch.randn(2 * B, 77, 768, device="cuda").bfloat16()
    with torch.no_grad():
        unet(x, t, ctx)                                        # warm (derived weights, hoisted context projections)
        calls = OrderedDict()
        depth = [0]
        orig = {}

        def wrap(name):
            f = getattr(E.UceHandle, name)
            orig[name] = f

            def g(self, *a, **k):
                if depth[0] == 0:
                    key = (name, tuple(sig_of(v) for v in a), tuple(sorted((kk, sig_of(vv)) for kk, vv in k.items())))
                    calls[key] = calls.get(key, 0) + 1
                depth[0] += 1
                try:
                    return f(self, *a, **k)
                finally:
                    depth[0] -= 1
            setattr(E.UceHandle, name, g)

        for m in METHODS:
            if hasattr(E.UceHandle, m):
                wrap(m)
        unet(x, t, ctx)
        for m, f in orig.items():
            setattr(E.UceHandle, m, f)
        torch.cuda.synchronize()
        # the whole call, eager and replayed from a hipGraph
        eager_us = timeit(lambda: unet(x, t, ctx), 5)
        H = E.UceHandle.get("cuda:0")
        rows = []
        for (name, a, k), cnt in calls.items():
            args = [rebuild(s) for s in a]
            kw = {kk: rebuild(vv) for kk, vv in k}
            fn = getattr(H, name)
            try:
                us = timeit(lambda: fn(*args, **kw))
            except Exception as err:  # noqa: BLE001
                us = float("nan")
                print(json.dumps({"name": name, "error": str(err)[:200]}), flush=True)
            gf = flops(name, a, dict(k)) * 1e-9
            shapes = [s[1] for s in a if s[0] == "T"]
            extra = {kk: (vv[1] if vv[0] != "T" else list(vv[1])) for kk, vv in k if vv[0] != "V" or vv[1] not in (None, False)}
            rows.append({"name": name, "shapes": [list(s) for s in shapes[:3]], "kw": extra, "count": cnt, "us": round(us, 2),
                         "step_us": round(us * cnt, 1), "gflop": round(gf, 2), "TFs": round(gf / us * 1e-3, 1) if gf else None})
    rows.sort(key=lambda r: -r["step_us"])
    total = sum(r["step_us"] for r in rows)
    by = {}
    for r in rows:
        by[r["name"]] = by.get(r["name"], 0.0) + r["step_us"]
    for r in rows[:top]:
        print(json.dumps(r), flush=True)
    print(json.dumps({"B": B, "distinct_calls": len(rows), "launches_counted": sum(r["count"] for r in rows),
                      "sum_isolated_us": round(total, 1), "unet_eager_us": round(eager_us, 1),
                      "gflop": round(sum(r["gflop"] * r["count"] for r in rows), 1),
                      "by_method_us": {k: round(v, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}), flush=True)


if __name__ == "__main__":
    main()
