"""Where does uce_edit's project + update form (one pass over W per 128 edit concepts) beat Delta + the dense apply?
Times uce_edit over a grid of concept counts under whatever UCE_SPLIT_MAX_NE / UCE_SPLIT_MAX_N the environment sets.
    UCE_SPLIT_MAX_N=128 python tools/ab_split.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import uce_oracle as O  # noqa: E402  (inputs only)
from uce_amd import edit as E  # noqa: E402

H = E.UceHandle.get("cuda:0")
d, rows = 768, 24960
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
W = dev(O.linear_default_weight(rows, d, np.random.default_rng(0)))
out = torch.empty_like(W)
res = []
for N_e, N_p in [(20, 120), (64, 100), (64, 300), (100, 60), (100, 300), (128, 100), (128, 500), (129, 0), (160, 40), (200, 100),
                 (256, 0), (256, 200)]:
    N = N_e + N_p
    Call = O.clip_like_embeddings(N + 1, d, seed=1)
    C, G = dev(Call[:N]), dev(np.repeat(Call[-1:], N_e, axis=0))
    s = dev(np.ones(N, np.float32))
    for _ in range(8):
        H.edit(C, G, s, 0.5, W, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        H.edit(C, G, s, 0.5, W, out=out)
    e1.record()
    torch.cuda.synchronize()
    H.status()
    res.append("%d+%d: %.4f" % (N_e, N_p, e0.elapsed_time(e1) / 40))
print(os.environ.get("UCE_SPLIT_MAX_NE"), os.environ.get("UCE_SPLIT_MAX_N"), " | ".join(res))
