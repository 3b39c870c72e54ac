"""GPU probe: latency of the Gram + Cholesky rider chain: uce_edit on a slab so small that the GEMM / update
kernels are negligible (the projection launch then lasts as long as its riders)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uce_amd import edit as E, synth
H = E.UceHandle.get("cuda:0")
d = 768
for N in (5, 50, 64, 100, 128):
    for rows in (1024, 24960):
        C = torch.from_numpy(synth.clip_like_embeddings(N, d, 0)).cuda(); G = C.roll(1, 0).contiguous(); s = torch.ones(N, device="cuda")
        W = torch.randn(rows, d, device="cuda") * 0.03; out = torch.empty_like(W)
        for _ in range(5): H.edit(C, G, s, 0.5, W, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): H.edit(C, G, s, 0.5, W, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"N={N:4d} rows={rows:6d}: {e0.elapsed_time(e1) * 5:7.2f} us per edit", flush=True)
