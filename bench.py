#!/usr/bin/env python3
"""Headline benchmark: concepts/s of the closed-form UCE edit of ALL SD-1.4 cross-attention K/V
projections (BASELINE.json configs[1]: erase 50 concepts, d = 768, 32 modules = one 24960 x 768
fp32 slab), inputs resident in HBM.  One "step" = one full pass of the hot path
(Gram -> SPD solve -> weight update for every module) = one `uce_edit` call.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The edit does not shard ("replicas only"): with N > 1 every rank edits its own
replica and `value` is the aggregate over ranks.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3     # dense f32-input MFMA peak
BF16_MFMA_PEAK_TF = 2500.0    # dense bf16 MFMA peak (no sparsity)

WORKLOADS = {
    # name: (N_edit, N_preserve, d, module table)
    "sd14_erase50": (50, 0, 768, "sd14"),
    "sd14_erase2p3": (2, 3, 768, "sd14"),
    "sd14_erase100": (100, 0, 768, "sd14"),
    "sd14_erase1000p500": (1000, 500, 768, "sd14"),
    "sdxl_debias36x2": (36, 0, 2048, "sdxl"),
}


def make_inputs(name: str, device):
    from uce_amd import synth as O
    n_e, n_p, d, table = WORKLOADS[name]
    mods = O.sd14_module_table() if table == "sd14" else O.sdxl_module_table()
    rows = sum(o for _, o in mods)
    N = n_e + n_p
    emb = O.clip_like_embeddings(N + 1, d, seed=0)
    rng = np.random.Generator(np.random.PCG64(0))
    bound = 1.0 / np.sqrt(d)
    W = torch.from_numpy(rng.uniform(-bound, bound, size=(rows, d)).astype(np.float32)).to(device)
    C = torch.from_numpy(emb[:N]).to(device)
    G = torch.from_numpy(np.repeat(emb[N:N + 1], n_e, axis=0)).to(device)
    s = torch.ones(N, dtype=torch.float32, device=device)
    return dict(C=C, G=G, s=s, W=W, mods=mods, rows=rows, d=d, n_e=n_e, n_p=n_p, emb=emb)


def _log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(inp, budget_s: float = 12.0):
    """The oracle (the reference's own op order on torch CPU: sequential rank-1 fp32 updates,
    torch.inverse and a GEMM per module) timed on the host cores.  Bounded sample: modules are
    timed one at a time (round-robin over the three width classes) until ~budget_s seconds are
    spent; the whole-edit time is the sum over all modules of their class's mean time."""
    from oracle import uce_oracle as O
    Wc = inp["W"].cpu()
    ws, off = [], 0
    for _, o in inp["mods"]:
        ws.append(Wc[off:off + o])
        off += o
    C, G = inp["C"].cpu(), inp["G"].cpu()
    n_e, n_p = inp["n_e"], inp["n_p"]
    edit = [C[i:i + 1] for i in range(n_e)]
    guide = [G[i:i + 1] for i in range(n_e)]
    pres = [C[n_e + i:n_e + i + 1] for i in range(n_p)]

    def one(w):
        t = time.perf_counter()
        O.uce_edit_ref([w], edit, guide, pres, 1.0, 1.0, 0.5)
        return time.perf_counter() - t

    # thread count: the reference would run with torch's default; many tiny ops do not scale
    # to a 100+ core host, so calibrate on one module and keep the fastest setting
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (4, 8, 16, 32) if c <= ncpu})
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        one(ws[0])
        t = one(ws[0])
        _log(f"cpu baseline calibration: {c} threads -> {t * 1e3:.1f} ms / module")
        if t < best_t:
            best, best_t = c, t
    torch.set_num_threads(best)
    classes = {}
    for i, w in enumerate(ws):
        classes.setdefault(w.shape[0], []).append(i)
    times = {k: [] for k in classes}
    t0, rr = time.perf_counter(), 0
    keys = sorted(classes)
    while time.perf_counter() - t0 < budget_s or min(len(v) for v in times.values()) < 1:
        k = keys[rr % len(keys)]
        idx = classes[k][(rr // len(keys)) % len(classes[k])]
        times[k].append(one(ws[idx]))
        rr += 1
        if rr >= 30 * len(ws):
            break
    el = time.perf_counter() - t0
    whole = sum(len(classes[k]) * (sum(times[k]) / len(times[k])) for k in keys)
    n = n_e + n_p
    return dict(value=round(n / whole, 2), unit="concepts/s", cores=best, kind="port",
                sample=f"{rr} single-module edits ({', '.join(f'{len(times[k])}x o={k}' for k in keys)}) of the "
                       f"{n}-concept workload in {el:.1f} s on {best} of {ncpu} host threads; whole edit "
                       f"({len(ws)} modules) = {whole:.3f} s extrapolated per width class")


def generation_leg(device, world, n_images, steps, edited_slab, inp, batch=8):
    """Secondary figure (BASELINE.json's second metric): images/s of the edited SD-1.4 pipeline,
    512x512, `steps` PNDM steps (+1 U-Net call), guidance 7.5, bf16, CPU-seeded latents, synthetic
    (seeded-random) weights, cross-attention through uce_xattn_fwd.  Every rank generates its own
    `n_images` prompts (the sharded rows of evalscripts/generate-images-sd.py); the edited attn2
    weights are broadcast from rank 0 first when world > 1."""
    from uce_amd.sd import pipeline as sdp
    from uce_amd import edit as E
    _log("generation leg: building the synthetic SD-1.4 pipeline")
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.bfloat16, device, synthetic=True, vae=True)
    if edited_slab is not None:
        blob = edited_slab.clone()
        if world > 1:
            torch.distributed.broadcast(blob, src=0)          # RCCL over xGMI: the one exchange step
        mods = E.collect_uce_modules(pipe.unet)
        off = 0
        state = {}
        for n, m in mods:
            r = m.weight.shape[0]
            state[n + ".weight"] = blob[off:off + r]
            off += r
        sdp.patch_unet(pipe, state)
    rank = int(os.environ.get("RANK", "0"))
    batch = max(1, min(batch, n_images))

    def run(first, count, nsteps):
        prompts = [f"synthetic prompt {rank * n_images + first + j}" for j in range(count)]
        gens = [torch.Generator().manual_seed(1000 + rank * n_images + first + j) for j in range(count)]
        return pipe(prompts if count > 1 else prompts[0], num_inference_steps=nsteps, guidance_scale=7.5,
                    generator=gens if count > 1 else gens[0])

    chunks = [(lo, min(batch, n_images - lo)) for lo in range(0, n_images, batch)]
    # A rank that fails must still reach every collective (the others would wait for it until the RCCL
    # time-out): errors are recorded, the barriers are always passed, and the flag is reduced at the end.
    failure = None
    try:
        for count in sorted({c for _, c in chunks}):              # warm-up: solver search, hipGraph capture
            run(0, count, 2)
    except Exception as err:  # noqa: BLE001
        failure = f"warm-up: {err!r}"
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if failure is None:
        try:
            for lo, count in chunks:
                run(lo, count, steps)                             # -> PIL images on the host, as pipe(...).images
        except Exception as err:  # noqa: BLE001
            failure = f"timed loop: {err!r}"
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el, 0.0 if failure is None else 1.0], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t[0].item())
        if failure is None and t[1].item() > 0:
            failure = "another rank failed"
    if failure is not None:
        _log("generation leg failed: " + failure)
        return {"metric": "images/sec 512x512 50-step", "value": None, "error": failure, "n_gpus": world}
    return {"metric": "images/sec 512x512 50-step", "value": round(world * n_images / el, 4), "unit": "images/s",
            "n_gpus": world, "images_per_rank": n_images, "prompts_per_unet_call": batch, "steps": steps, "dtype": "bf16",
            "scaling": "weak",
            "data": "synthetic weights, synthetic prompts, CPU-seeded latents", "seconds": round(el, 3)}


def xattn_leg(device, batches=(2, 16)):
    """Cross-attention kernel alone at SD-1.4's four attn2 shapes (H = 8, Lk = 77, bf16) at B = 2 (the CFG pair
    of one prompt) and at the batch the generation leg runs (2 x prompts per U-Net call): algorithmic bytes =
    Q + O + K + V once, per launch, vs the HBM peak."""
    from uce_amd import edit as E
    H = E.UceHandle.get(device)
    out = []
    for B in batches:
        for Lq, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
            C = 8 * dh
            q = torch.randn(B, Lq, C, device=device).bfloat16()
            k = torch.randn(B, 77, C, device=device).bfloat16()
            v = torch.randn_like(k)
            o = torch.empty_like(q)
            ms = time_kernel(lambda: H.xattn(q, k, v, 8, out=o), 100)
            byts = 2.0 * (B * Lq * C * 2) + 2.0 * (B * 77 * C * 2)
            out.append({"B": B, "Lq": Lq, "dh": dh, "avg_us": round(ms * 1e3, 2), "bytes": byts,
                        "achieved_GBs": round(byts / (ms * 1e-3) / 1e9, 1),
                        "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return {"kernel": "k_xattn", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "shapes": out,
            "note": "B = 2 launches move 1.4-10.7 MB each (launch/latency-bound); the batched rows are the ones "
                    "the generation leg issues"}


def sattn_leg(device, B):
    """Self-attention kernel (attn1 of the U-Net, uce_sattn_fwd) at SD-1.4's four shapes and the generation batch,
    beside torch's scaled_dot_product_attention on the same tensors; 4*B*H*L^2*dh flop per call vs the dense bf16
    MFMA peak (the loop is VALU-bound on the online softmax, not MFMA-bound)."""
    import torch.nn.functional as F
    from uce_amd import edit as E
    H = E.UceHandle.get(device)
    out = []
    for L, dh in ((4096, 40), (1024, 80), (256, 160), (64, 160)):
        C = 8 * dh
        q = torch.randn(B, L, C, device=device).bfloat16()
        k, v = torch.randn_like(q), torch.randn_like(q)
        o = torch.empty_like(q)
        ms = time_kernel(lambda: H.sattn(q, k, v, 8, out=o), 10)
        sp = lambda t: t.view(B, L, 8, dh).transpose(1, 2)
        ms_t = time_kernel(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)), 10)
        fl = 4.0 * B * 8 * L * L * dh
        out.append({"B": B, "L": L, "dh": dh, "avg_us": round(ms * 1e3, 1), "achieved_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1),
                    "frac": round(fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4), "torch_sdpa_us": round(ms_t * 1e3, 1)})
    return {"kernel": "k_sattn (+ k_vt)", "bound": "mfma", "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "shapes": out}


def time_kernel(fn, iters: int):
    """Average duration (ms) of `fn`'s launches on the current stream, HIP events around `iters`
    back-to-back launches (the library enqueues on torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="sd14_erase50", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", default="auto", choices=["auto", "primal", "dual"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-images", type=int, default=32,
                    help="images per rank for the secondary images/s figure (0 = skip)")
    ap.add_argument("--gen-batch", type=int, default=16, help="prompts denoised per U-Net call")
    ap.add_argument("--gen-steps", type=int, default=50)
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from uce_amd import edit as E
    from uce_amd import cli
    H = E.UceHandle.get(device)
    inp = make_inputs(args.workload, device)
    C, G, s, W = inp["C"], inp["G"], inp["s"], inp["W"]
    N, d, rows, n_e = C.shape[0], inp["d"], inp["rows"], inp["n_e"]
    algo = cli.ALGO_IDS[args.algo]
    out = torch.empty_like(W)
    H.reserve(d, max(N, d))
    H.reserve_rows(rows, max(n_e, 1))

    def step():
        H.edit(C, G, s, 0.5, W, out=out, algo=algo)

    _log(f"inputs ready: N={N} d={d} rows={rows}; warm-up")
    for _ in range(args.warmup):
        step()
    H.status()
    _log("timed region")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    H.status()

    _log(f"timed region done: {1e3 * elapsed / args.steps:.4f} ms/step")
    # ---- per-kernel timing of the dominant kernel, HIP events on the launch stream
    use_dual = (algo == 2) or (algo == 0 and ((N + 63) // 64) * 64 < d)
    iters = max(20, min(200, args.steps))
    if use_dual and 1 <= n_e <= 256 and d in (768, 1024, 2048) and rows >= 1024:
        # uce_edit's path here: projection (+ riders) -> triangular solves -> update; the update is the
        # HBM-bound pass over the weights and the longest kernel
        Dm, R = H.dual_factors(C, G, s, 0.5)
        T = H.lowrank_project(W, Dm)
        ms = time_kernel(lambda: H.lowrank_update(W, T, R, out=out), iters)
        nep = T.shape[1]
        alg_bytes = 8.0 * rows * d + 4.0 * rows * nep + 4.0 * n_e * d      # W in + W out, T in, R once
        roof = dict(kernel="k_lr_update_s" if n_e <= 128 else "k_lr_update", bound="hbm",
                    achieved=round(alg_bytes / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    avg_ms=round(ms, 5), algorithmic_bytes=alg_bytes)
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        ms_p = time_kernel(lambda: H.lowrank_project(W, Dm), iters)
        flops_p = 2.0 * rows * d * nep
        roof["second_kernel"] = dict(kernel="k_lr_project", bound="mfma", achieved=round(flops_p / (ms_p * 1e-3) / 1e12, 2),
                                     peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", avg_ms=round(ms_p, 5),
                                     frac=round(flops_p / (ms_p * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4),
                                     note="timed alone, without the Gram+Cholesky rider blocks it carries inside uce_edit")
    elif use_dual and n_e <= 256:
        Dm, R = H.dual_factors(C, G, s, 0.5)
        ms = time_kernel(lambda: H.apply_lowrank(W, Dm, R, out=out), iters)
        alg_bytes = 8.0 * rows * d + 8.0 * n_e * d          # W in + W out (+ the two factors once)
        roof = dict(kernel="k_apply_lowrank", bound="hbm", achieved=round(alg_bytes / (ms * 1e-3) / 1e9, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", avg_ms=round(ms, 5), algorithmic_bytes=alg_bytes)
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    else:
        A, Bt = H.gram(C, G, s, 0.5)
        DT = H.solve_delta(A, Bt)
        ms = time_kernel(lambda: H.apply(W, DT, out=out), iters)
        flops = 2.0 * rows * d * d
        roof = dict(kernel="k_apply", bound="mfma", achieved=round(flops / (ms * 1e-3) / 1e12, 2),
                    peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", avg_ms=round(ms, 5), algorithmic_flops=flops)
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    roof["traffic"] = None
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file)).get(args.workload, {}).get(roof["kernel"])
            roof["traffic"] = t["total_bytes"] if isinstance(t, dict) else t
            roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (profiles/traffic.json), bytes per launch"
        except Exception:
            pass

    gen = None
    if args.gen_images > 0:
        gen = generation_leg(device, world, args.gen_images, args.gen_steps, out if out.shape[1] == 768 else None, inp,
                             args.gen_batch)
    result = {
        "metric": "concepts/sec closed-form edit (SD-1.4, 768-d)",
        "value": round(world * N * args.steps / elapsed, 1),
        "unit": "concepts/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (f64 Gram/solve)",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_e} erase + {inp['n_p']} preserve concepts, d={d}, "
                               f"{len(inp['mods'])} attn2 to_k/to_v modules = one {rows}x{d} fp32 slab, "
                               f"lambda 0.5, algo {args.algo}",
                   "parallelism": "replicas only" if world > 1 else "single GPU"},
        "roofline": roof,
    }
    if gen is not None:
        result["generate"] = gen
    if rank == 0 and args.gen_images > 0:
        gb = 2 * max(1, min(args.gen_batch, args.gen_images))
        result["xattn"] = xattn_leg(device, (2, gb))
        result["sattn"] = sattn_leg(device, gb)
    if rank == 0:
        _log("gpu part: " + json.dumps(result))
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(inp, args.cpu_budget)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
