"""GPU stress of the primal path's persistent Cholesky launch with its rider jobs (f16 split of W_old, the Bt tiles) and
of the f16 dense apply: `iters` edits with random concept counts / row counts, alternately on two handles and two streams
without synchronisation in between, every result against torch fp64.
    python tools/stress_primal.py [iters] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import uce_oracle as O  # noqa: E402  (checker only)
from uce_amd import edit as E  # noqa: E402
from uce_amd import lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.Generator(np.random.PCG64(seed))
H1, H2 = E.UceHandle.get("cuda:0"), E.UceHandle("cuda:0")
side = torch.cuda.Stream()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
jobs = []
torch.cuda.synchronize()
for it in range(iters):
    d = int(rng.choice([256, 512, 768, 768, 768, 1024]))
    N_e = int(rng.integers(1, 900))
    N = N_e + int(rng.integers(0, 700))
    rows = int(rng.integers(64, 6000))
    Call = O.clip_like_embeddings(N + 1, d, seed=int(rng.integers(1 << 30)))
    C, G = dev(Call[:N]), dev(np.repeat(Call[N:N + 1], N_e, axis=0))
    s = dev((0.5 + rng.random(N)).astype(np.float32))
    W = dev(O.linear_default_weight(rows, d, rng))
    if it % 2 == 0:
        out = H1.edit(C, G, s, 0.5, W, algo=L.ALGO_PRIMAL)
    else:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = H2.edit(C, G, s, 0.5, W, algo=L.ALGO_PRIMAL)
    jobs.append((C, G, s, W, N_e, d, out))
torch.cuda.synchronize()
H1.status()
H2.status()
worst = 0.0
for it, (C, G, s, W, N_e, d, out) in enumerate(jobs):
    C64, s64, W64 = C.double(), s.double(), W.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda") + C64.T @ (s64[:, None] * C64)
    Delta = torch.linalg.solve(A, (s64[:N_e, None] * C64[:N_e]).T @ (G - C[:N_e]).double()).T
    want = W64 + W64 @ Delta
    err = ((out.double() - want).norm() / want.norm()).item()
    worst = max(worst, err)
    if not err < 1e-5:
        print(f"FAIL job {it}: d={d} N={C.shape[0]} N_e={N_e} rows={W.shape[0]} relF={err:.3e}")
        sys.exit(1)
H2.close()
print(f"{iters} primal edits on two handles / streams, worst relF vs fp64 {worst:.2e}")
