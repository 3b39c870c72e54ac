"""uce_sattn_packed_fwd against uce_sattn_packed_exp2_fwd (q pre-scaled by the projection's epilogue) at the generation batch, and the
projection itself with and without the column scale: us per launch, error against fp64 on one (batch, head) slice.
   python tools/probe_sattn_exp2.py [B]"""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from uce_amd import edit as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
H = E.UceHandle.get(dev)
heads, dh, L = 8, 40, 4096
C = heads * dh
c = dh ** -0.5 * 1.4426950408889634
g = torch.Generator(device="cuda").manual_seed(1)
out = {"B": B, "L": L, "dh": dh}
for gain, tag in ((1.0, "exchangeable"), (5.0, "peaked_x5")):
    f = torch.randn(B, L, 3 * C, device=dev, generator=g)
    f[..., :2 * C] *= gain
    plain = f.bfloat16()
    f[..., :C] *= c
    scaled = f.bfloat16()
    del f
    t0 = bench.time_kernel(lambda: H.sattn_packed(plain, heads), 6) * 1e3
    t1 = bench.time_kernel(lambda: H.sattn_packed_exp2(scaled, heads), 6) * 1e3
    o0, o1 = H.sattn_packed(plain, heads), H.sattn_packed_exp2(scaled, heads)
    # fp64 on batch 0
    def ref(qkv, ln2):
        sp = lambda t: t[:1].double().view(1, L, heads, dh).transpose(1, 2)
        q, k, v = (sp(t) for t in qkv.split(C, dim=-1))
        s = q @ k.transpose(-1, -2) * ln2
        return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(1, L, C)
    r0, r1 = ref(plain, dh ** -0.5), ref(scaled, 0.6931471805599453)
    rf = lambda a, b: float((a.double() - b).norm() / b.norm())
    out[tag] = {"packed_us": round(t0, 1), "exp2_us": round(t1, 1), "packed_err_fp64": rf(o0[:1], r0), "exp2_err_fp64": rf(o1[:1], r1),
                "exp2_vs_unscaled_fp64": rf(o1[:1], r0)}
    del plain, scaled, o0, o1
x = torch.randn(B * L, C, device=dev, generator=g).bfloat16()
w = (torch.randn(3 * C, C, device=dev, generator=g) * 0.05).bfloat16()
out["projection_us"] = round(bench.time_kernel(lambda: H.linear(x, w), 6) * 1e3, 1)
out["projection_colscale_us"] = round(bench.time_kernel(lambda: H.linear_colscale(x, w, C, c), 6) * 1e3, 1)
print(json.dumps(out))
