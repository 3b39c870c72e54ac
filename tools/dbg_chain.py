"""Phase timeline of the rider chain inside the projection launch (Gram riders -> factorising block -> solve riders
beside the projection blocks).  Needs a library built with UCE_CHAIN_DEBUG=1 (python -m uce_amd.build):

    UCE_CHAIN_DEBUG=1 python -m uce_amd.build && python tools/dbg_chain.py [workload]

Prints, per workgroup, the wall-clock stamps (us since the first block started) of its phases:
  Gram rider   : start | Gram MFMAs done | slab published, ticket drawn | (last arriver) slabs summed | factor done | announced
  solve rider  : start | C tile loaded | factorisation seen | L^-1 loaded | products done | R stored
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
buf = (ctypes.c_ulonglong * (64 * 16))()
L = lib.load()
L.uce_debug_read.argtypes = [ctypes.c_void_p]
assert L.uce_debug_read(buf) == 0
a = np.array(buf[:]).reshape(64, 16).astype(np.int64)
t0 = a[:, 0][a[:, 0] > 0].min()
for b in range(64):
    row = [(int(x) - int(t0)) / 100.0 if x > 0 else None for x in a[b, :6]]
    if b < 20 or b % 16 == 0:
        cyc = [int(a[b, 8 + k + 1]) - int(a[b, 8 + k]) if a[b, k + 1] > 0 and a[b, k] > 0 else None for k in range(5)]
        print(b, row, "shader-clock cycles per phase:", cyc)

