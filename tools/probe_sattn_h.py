"""k_sattn_h (UCE_SATTN_QT=4) against the by-shape default: parity vs fp64 and time per launch.  GPU box only."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E


def handle(qt, vti="0"):
    os.environ["UCE_SATTN_QT"] = qt
    os.environ["UCE_SATTN_VTI"] = vti
    h = E.UceHandle("cuda:0")
    del os.environ["UCE_SATTN_QT"], os.environ["UCE_SATTN_VTI"]
    return h


def ref(q, k, v, heads):
    B, Lq, C = q.shape
    dh = C // heads
    sp = lambda t: t.double().view(B, t.shape[1], heads, dh).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5
    return (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(B, Lq, C)


def relf(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = {}
hs = {"def": handle("0"), "h": handle("4"), "h_vti": handle("4", "1"), "h_pre": handle("4", "2")}
g = torch.Generator().manual_seed(3)
for (B, Hh, Lq, Lk, dh, dt) in [(2, 8, 4096, 4096, 40, torch.bfloat16), (2, 8, 1024, 1024, 40, torch.bfloat16), (1, 8, 300, 130, 40, torch.bfloat16),
                                (3, 4, 77, 64, 40, torch.float16), (2, 4, 100, 33, 40, torch.bfloat16), (2, 4, 100, 97, 40, torch.bfloat16),
                                (1, 4, 257, 128, 40, torch.bfloat16), (1, 2, 40, 1, 40, torch.bfloat16), (2, 3, 500, 1000, 32, torch.float16),
                                (1, 8, 256, 4000, 48, torch.bfloat16)]:
    C = Hh * dh
    q = torch.randn(B, Lq, C, generator=g).to(dt).cuda()
    k = torch.randn(B, Lk, C, generator=g).to(dt).cuda()
    v = torch.randn(B, Lk, C, generator=g).to(dt).cuda()
    r = ref(q, k, v, Hh)
    row = {n: relf(h.sattn(q, k, v, Hh), r) for n, h in hs.items()}
    row["rep"] = bool(torch.equal(hs["h"].sattn(q, k, v, Hh), hs["h"].sattn(q, k, v, Hh)))
    out[f"{B}x{Hh}x{Lq}x{Lk}x{dh}{'h' if dt == torch.float16 else 'b'}"] = row
    print(B, Hh, Lq, Lk, dh, dt, row, flush=True)
# rising max
q = (torch.randn(1, 128, 160, generator=g).abs() * 3).bfloat16().cuda()
k = (torch.randn(1, 640, 160, generator=g).abs() * torch.linspace(0.1, 3.0, 640)[None, :, None]).bfloat16().cuda()
v = torch.randn(1, 640, 160, generator=g).bfloat16().cuda()
print("rising", {n: relf(h.sattn(q, k, v, 4), ref(q, k, v, 4)) for n, h in hs.items()}, flush=True)
# packed + time
for B, L, Hh, dh in [(32, 4096, 8, 40), (128, 4096, 8, 40), (32, 1024, 8, 40), (128, 1024, 8, 40), (128, 256, 8, 40)]:
    C = Hh * dh
    qkv = torch.randn(B, L, 3 * C, generator=g).bfloat16().cuda() if B * L <= 32 * 4096 else (torch.randn(B, L, 3 * C, device="cuda") ).bfloat16()
    o = {n: h.sattn_packed(qkv, Hh) for n, h in hs.items()}
    row = {n: relf(o[n], o["def"]) for n in hs}
    t = {n: timeit(lambda h=h: h.sattn_packed(qkv, Hh)) for n, h in hs.items()}
    fl = 4.0 * B * Hh * L * L * dh
    print("packed", B, L, row, {n: round(x, 1) for n, x in t.items()}, {n: round(fl / x / 1e6, 1) for n, x in t.items()}, "TF/s", flush=True)
    out[f"packed{B}x{L}"] = dict(us=t, tf={n: fl / x / 1e6 for n, x in t.items()})
json.dump(out, open(os.path.join(os.environ.get("OUT", "gpurun_out"), "sattn_h.json"), "w"), indent=1)
