"""Phase timeline of the one-launch register-resident edit (uce_edit_resident.hip).  Needs a library built with UCE_CHAIN_DEBUG=1:

    UCE_CHAIN_DEBUG=1 python -m uce_amd.build && UCE_CHAIN_DEBUG=1 python tools/dbg_resident.py [workload]

Per workgroup (blocks 0-63), us since the first block started.  Riders as tools/dbg_chain.py prints them; D-prep: start | done;
main: start | W tile loaded + row maxima | D fragments ready | phase A done | T split | stage 4 (R complete) seen | last store
issued | stores drained."""
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
for _ in range(20):
    H.edit(inp["C"], inp["G"], inp["s"], 0.5, inp["W"], out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
L = lib.load()
L.uce_debug_read_res.argtypes = [ctypes.c_void_p]
assert L.uce_debug_read_res(buf) == 0
a = np.array(buf[:]).reshape(64, 32).astype(np.int64)
starts = np.concatenate([a[:, 0][a[:, 0] > 0], a[:, 8][a[:, 8] > 0]])
t0 = starts.min()
for b in range(64):
    for role, base in (("gram|dprep|main", 0), ("solve", 8)):
        if a[b, base] <= 0:
            continue
        row = [round((int(x) - int(t0)) / 100.0, 2) if x > 0 else None for x in a[b, base:base + 8]]
        print(b, role, row)
    if a[b, 8] > 0 and a[b, 6] > 0 and b >= 29:        # main: per-stage stamps of phase B (stages 0-7) and shader-clock deltas
        st = [round((int(x) - int(t0)) / 100.0, 2) for x in a[b, 8:16]]
        ck = [int(a[b, 16 + 8 + k + 1]) - int(a[b, 16 + 8 + k]) for k in range(7)]
        print(b, "phase B stages", st, "cycles/stage", ck)
