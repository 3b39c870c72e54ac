"""Phase timeline of the rider chain inside the projection launch (Gram riders -> factorising block -> solve riders
beside the projection blocks).  Needs a library built with UCE_CHAIN_DEBUG=1 (python -m uce_amd.build):

    UCE_CHAIN_DEBUG=1 python -m uce_amd.build && python tools/dbg_chain.py [workload]

Prints, per workgroup, the wall-clock stamps (us since the first block started) of its phases:
  Gram rider   : start | Gram MFMAs done | slab published, ticket drawn | (last arriver) slabs summed |
                 one block: factor done | stage 2 published;  two blocks: L_00^-1 done + stage 1 | stage 2 published
  solve rider  : start | C tiles loaded | one block: stage 2 seen, two blocks: Y0 done | factor blocks loaded |
                 products done | R stored          (a Gram rider that did not draw the last ticket goes on as a solve rider)
  projection   : start | end"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from uce_amd import edit as E, lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "sd14_erase50"
H = E.UceHandle.get("cuda:0")
inp = bench.make_inputs(wl, "cuda:0")
out = torch.empty_like(inp["W"])
H.reserve(inp["d"], max(inp["d"], inp["C"].shape[0]))
H.reserve_rows(inp["rows"], max(inp["n_e"], 1))
for _ in range(20):
    H.edit(inp["C"], inp["G"], inp["s"], 0.5, inp["W"], out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
L = lib.load()
L.uce_debug_read.argtypes = [ctypes.c_void_p]
assert L.uce_debug_read(buf) == 0
a = np.array(buf[:]).reshape(64, 32).astype(np.int64)
starts = np.concatenate([a[:, 0][a[:, 0] > 0], a[:, 8][a[:, 8] > 0]])
t0 = starts.min()
for b in range(64):
    if not (b < 40 or b % 16 == 0):
        continue
    for role, base in (("gram/proj", 0), ("solve", 8)):
        if a[b, base] <= 0:
            continue
        row = [round((int(x) - int(t0)) / 100.0, 2) if x > 0 else None for x in a[b, base:base + 6]]
        cyc = [int(a[b, 16 + base + k + 1]) - int(a[b, 16 + base + k]) if a[b, base + k + 1] > 0 and a[b, base + k] > 0 else None
               for k in range(5)]
        print(b, role, row, "cycles/phase:", cyc)

