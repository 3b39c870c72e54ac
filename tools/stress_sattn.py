"""GPU stress of the self-attention kernels: random (batch, heads, head dim, query / key lengths, dtype) - ragged against every tile
size - through uce_sattn_fwd, and where Lq = Lk through uce_sattn_packed_fwd and uce_sattn_packed_exp2_fwd (q pre-scaled), on the
by-rule handle and on one that forces the two-tile kernel (UCE_SATTN_QT=4: k_sattn_h and its exp2-domain form at every shape with
dh <= 48); every result against fp64 on the GPU.
    python tools/stress_sattn.py [iters] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from uce_amd import edit as E  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
H0 = E.UceHandle.get("cuda:0")
os.environ["UCE_SATTN_QT"] = "4"
H4 = E.UceHandle("cuda:0")
del os.environ["UCE_SATTN_QT"]
TOL = {torch.bfloat16: 8e-3, torch.float16: 1.5e-3}
LOG2E = 1.4426950408889634


def ref(q, k, v, heads, scale):
    B, Lq, C = q.shape
    dh = C // heads
    sp = lambda t: t.double().view(B, t.shape[1], heads, dh).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) * scale
    return (torch.softmax(s, dim=-1) @ sp(v)).transpose(1, 2).reshape(B, Lq, C)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


worst, n_exp2 = 0.0, 0
g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
for it in range(iters):
    dh = int(rng.choice([16, 40, 40, 48, 64, 80, 80, 96, 128, 160, 160]))
    heads = int(rng.integers(1, 11))
    B = int(rng.integers(1, 7))
    big = it % 7 == 0
    Lk = int(rng.integers(1, 2600 if big else 400))
    Lq = Lk if it % 2 == 0 else int(rng.integers(1, 1200 if big else 300))
    dtype = torch.float16 if it % 3 == 0 else torch.bfloat16
    gain = float(rng.choice([1.0, 1.0, 3.0]))
    C = heads * dh
    qf = torch.randn(B, Lq, C, device="cuda", generator=g) * gain
    kf = torch.randn(B, Lk, C, device="cuda", generator=g) * gain
    vf = torch.randn(B, Lk, C, device="cuda", generator=g)
    q, k, v = qf.to(dtype), kf.to(dtype), vf.to(dtype)
    want = ref(q, k, v, heads, dh ** -0.5)
    for name, Hx in (("rule", H0), ("qt4", H4)):
        o = Hx.sattn(q, k, v, heads)
        e = rel(o, want)
        ok = bool(torch.isfinite(o.float()).all()) and e < TOL[dtype]
        if Lq == Lk:
            qkv = torch.cat([q, k, v], dim=-1)
            o2 = Hx.sattn_packed(qkv, heads)
            ok = ok and torch.equal(o, o2)
            qs = (qf * (dh ** -0.5 * LOG2E)).to(dtype)
            o3 = Hx.sattn_packed_exp2(torch.cat([qs, k, v], dim=-1), heads)
            e3 = rel(o3, ref(qs, k, v, heads, 0.6931471805599453))
            n_exp2 += int(Hx.sattn_exp2_form(B, heads, Lq, dh))
            ok = ok and bool(torch.isfinite(o3.float()).all()) and e3 < TOL[dtype]
            e = max(e, e3)
        worst = max(worst, e / TOL[dtype])
        if not ok:
            print(f"FAIL it={it} {name} B={B} H={heads} dh={dh} Lq={Lq} Lk={Lk} {dtype} gain={gain} err={e:.3e}")
            sys.exit(1)
torch.cuda.synchronize()
H4.close()
print(f"ok: {iters} shapes x 2 handles, worst error / tolerance {worst:.3f}, {n_exp2} calls on the exp2-domain kernel")
