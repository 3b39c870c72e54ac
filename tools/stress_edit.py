"""GPU stress of the in-launch rider chain (Gram -> slabs -> 64 x 64 Cholesky inverse -> solve riders) and of the launch
chain: `iters` edits with random concept counts back to back on one handle, every result against torch fp64.
    python tools/stress_edit.py [iters] [seed] [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import uce_oracle as O  # noqa: E402  (checker only)
from uce_amd import edit as E  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.Generator(np.random.PCG64(seed))
H = E.UceHandle.get("cuda:0")
d, rows = 768, (int(sys.argv[3]) if len(sys.argv) > 3 else 4096)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
W = dev(O.linear_default_weight(rows, d, rng))
W64 = W.double()
worst = 0.0
for it in range(iters):
    N_e = int(rng.integers(1, 200 if it % 10 == 0 else 129))
    N_p = int(rng.integers(0, max(1, 129 - N_e))) if N_e < 128 else 0
    N = N_e + N_p
    Call = O.clip_like_embeddings(N + 1, d, seed=int(rng.integers(1 << 30)))
    C, G = dev(Call[:N]), dev(np.repeat(Call[N:N + 1], N_e, axis=0))
    s = dev((0.5 + rng.random(N)).astype(np.float32))
    out = H.edit(C, G, s, 0.5, W, check=True)
    C64, s64 = C.double(), s.double()
    A = 0.5 * torch.eye(d, dtype=torch.float64, device="cuda") + C64.T @ (s64[:, None] * C64)
    Delta = torch.linalg.solve(A, (s64[:N_e, None] * C64[:N_e]).T @ (G - C[:N_e]).double()).T
    err = ((out.double() - (W64 + W64 @ Delta)).norm() / (W64 + W64 @ Delta).norm()).item()
    worst = max(worst, err)
    if not err < 1e-5:
        print(f"FAIL iteration {it}: N_e={N_e} N_p={N_p} relF={err:.3e}")
        sys.exit(1)
print(f"{iters} edits, worst relF vs fp64 {worst:.2e}")
