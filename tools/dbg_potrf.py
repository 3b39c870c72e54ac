"""Phase timeline of the persistent Cholesky's walker (k_potrf_la, uce_solve.hip).  Needs the debug library:

    UCE_CHAIN_DEBUG=1 python -m uce_amd.build && UCE_CHAIN_DEBUG=1 python tools/dbg_potrf.py

Per diagonal block k, us since the walker started: block begins | tiles in LDS, L^-1 stores drained | L_k,k-1 formed |
Schur complement in place (factor starts) | factor done | side waves start polling for block k+1's tiles | those tiles fetched."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from uce_amd import edit as E, lib  # noqa: E402

H = E.UceHandle.get("cuda:0")
inp = bench.make_inputs("sd14_erase1000p500", "cuda:0")
out = torch.empty_like(inp["W"])
H.reserve(inp["d"], max(inp["d"], inp["C"].shape[0]))
for _ in range(10):
    H.edit(inp["C"], inp["G"], inp["s"], 0.5, inp["W"], out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (32 * 8))()
L = lib.load()
L.uce_debug_read_la.argtypes = [ctypes.c_void_p]
assert L.uce_debug_read_la(buf) == 0
a = np.array(buf[:]).reshape(32, 8).astype(np.int64)
t0 = a[0, 0]
names = ["begin", "tiles in", "L formed", "factor starts", "factor done", "side polls", "side fetched"]
for k in range(12):
    print(k, {n: round((int(a[k, i]) - int(t0)) / 100.0, 2) if a[k, i] > 0 else None for i, n in enumerate(names)})
