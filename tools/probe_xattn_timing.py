"""Why rocprofv3 says k_xattn_g<40> takes 137 us and bench.py's events 161-170 us for the same launch: the same shape
(B = 128, Lq = 4096, dh = 40, 77 keys) timed as bursts of 1 / 4 / 20 / 100 / 1000 back-to-back launches after an idle gap, with the
shader clock read before and after (rocm-smi), and each launch of a 100-burst bracketed on its own.  Usage (GPU box): python tools/probe_xattn_timing.py"""
import json
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uce_amd import edit as E  # noqa: E402


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        return {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower()}
    except Exception as err:  # noqa: BLE001
        return {"error": repr(err)[:80]}


def main():
    H = E.UceHandle.get("cuda:0")
    B, Lq, C = 128, 4096, 320
    q = torch.randn(B, Lq, C, device="cuda").bfloat16()
    k = torch.randn(B, 77, C, device="cuda").bfloat16()
    v = torch.randn_like(k)
    o = torch.empty_like(q)
    fn = lambda: H.xattn(q, k, v, 8, out=o)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    res = {"shape": [B, Lq, C], "bytes": 2.0 * B * Lq * C * 2 + 2.0 * B * 77 * C * 2}
    for burst in (1, 4, 20, 100, 1000):
        time.sleep(1.0)                                      # idle gap: the clock governor sees an idle chip
        c0 = sclk()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(burst):
            fn()
        e1.record()
        torch.cuda.synchronize()
        c1 = sclk()
        res[f"burst_{burst}_us"] = round(e0.elapsed_time(e1) / burst * 1e3, 1)
        res[f"burst_{burst}_clk"] = [c0, c1]
    time.sleep(1.0)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) * 1e3 for a, b in evs]
    res["each_of_100_us"] = {"first5": [round(x, 1) for x in per[:5]], "median": round(sorted(per)[50], 1), "last5": [round(x, 1) for x in per[-5:]]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
