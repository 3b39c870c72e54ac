"""GroupNorm (+ SiLU) at the generation batch: the two-kernel form against the one-launch form at every size (UCE_GN_FUSED=1 | 2), each in a
process of its own (the switch is read when the handle is created).  python tools/probe_gn.py"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import bench
    from uce_amd import edit as E
    H = E.UceHandle.get("cuda:0")
    out = {}
    for N, hw, C in ((256, 64, 320), (256, 64, 640), (256, 32, 640), (256, 32, 1280), (256, 16, 1280), (256, 16, 2560), (256, 8, 1280), (2, 64, 320)):
        x = torch.randn(N, C, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(C, device="cuda").to(torch.bfloat16)
        b = torch.randn(C, device="cuda").to(torch.bfloat16)
        y = H.groupnorm_nhwc(x, w, b, 32, 1e-5, True)
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float(), 32, w.float(), b.float(), 1e-5))
        err = float((y.float() - ref).norm() / ref.norm())
        ms = bench.time_kernel(lambda: H.groupnorm_nhwc(x, w, b, 32, 1e-5, True), 20)
        byts = 2.0 * x.numel() * 2
        out[f"N{N}_{hw}x{hw}_C{C}"] = {"us": round(ms * 1e3, 1), "GBs": round(byts / ms / 1e6, 0), "relF": err}
    print(json.dumps(out))
else:
    res = {}
    for v in ("1", "2"):
        r = subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, UCE_GN_FUSED=v), capture_output=True, text=True)
        res[v] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-800:]}
    for k in res["1"]:
        a, b = res["1"].get(k), res["2"].get(k) if isinstance(res["2"], dict) else None
        print(k, "two-kernel/rule", a, "| every size", b)
