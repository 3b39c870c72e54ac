"""A/B of uce_edit at the headline workload (50 concepts, SD-1.4 slab): the projection + update launch pair against the one-launch
form with 0 / 8 / 12 work items' rows of W_old held in registers across the wait for R (UCE_EDIT_FUSED).  Usage: python tools/ab_edit_fused.py [N]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uce_amd import edit as E  # noqa: E402


def handle(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return E.UceHandle("cuda:0")
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sd14_erase50"
    inp = bench.make_inputs(name, "cuda:0")
    out = {"workload": name}
    ref = None
    for mode in (0, 1, 8, 101, 108, 112, 0, 108):
        H = handle(UCE_EDIT_FUSED=mode)
        W_new = torch.empty_like(inp["W"])
        fn = lambda: H.edit(inp["C"], inp["G"], inp["s"], 0.5, inp["W"], out=W_new)
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        if ref is None:
            ref = W_new.clone()
        err = float((W_new.double() - ref.double()).norm() / ref.double().norm())
        out.setdefault(f"fused{mode}_us", []).append(round(us, 2))
        out[f"fused{mode}_rel_vs_two_launch"] = err
        H.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
