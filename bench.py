#!/usr/bin/env python3
"""Headline benchmark: concepts/s of the closed-form UCE edit of ALL SD-1.4 cross-attention K/V
projections (BASELINE.json configs[1]: erase 50 concepts, d = 768, 32 modules = one 24960 x 768
fp32 slab), inputs resident in HBM.  One "step" = one full pass of the hot path
(Gram -> SPD solve -> weight update for every module) = one `uce_edit` call.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script launches itself under
`torch.distributed.run` (one rank per GPU, RCCL); launched by torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE as usual.  The edit does not shard ("replicas only"):
every rank edits its own replica and `value` is the aggregate over ranks; the secondary
`generate` leg (images/s) does shard - rank r generates its own prompts after ONE RCCL broadcast of
the edited weights.  Prints ONE JSON line on rank 0.

The line carries, beside the driver's keys:
  roofline       the kernel of the step with the LARGEST time share (per-kernel durations measured live with HIP
                 events on the launch stream, uce_profile_begin/_end), every other kernel of the step under
                 `kernels`, and the step-level figures `step_frac` (roofline floor / measured step) and
                 `step_traffic_ratio` (PMC HBM bytes of all kernels / algorithmic bytes of the step)
  configs        the other single-GPU BASELINE configs (0: 2 erase + 3 preserve, 2: 1000 + 500, 3: the SDXL slab)
  generate       images/s of the edited pipeline (BASELINE's second metric), xattn / sattn: the attention kernels alone
  cpu_baseline   the reference's op order on the host cores (oracle, torch CPU)
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import tempfile
import socket
import statistics
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3     # dense f32-input MFMA peak
F64_MFMA_PEAK_TF = 78.6      # dense f64 MFMA peak
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 MFMA peak (no sparsity)
CPU_BASELINE_THREADS = 4     # first point of the thread sweep (the fastest count on the GPU box's EPYC host so far)

WORKLOADS = {
    # name: (N_edit, N_preserve, d, module table, BASELINE.json config index)
    "sd14_erase50": (50, 0, 768, "sd14", 1),
    "sd14_erase2p3": (2, 3, 768, "sd14", 0),
    "sd14_erase100": (100, 0, 768, "sd14", None),
    "sd14_erase1000p500": (1000, 500, 768, "sd14", 2),
    "sdxl_debias36x2": (36, 0, 2048, "sdxl", 3),
}
# legs of the one JSON line beside the headline workload: BASELINE configs 0, 2, 3 and the north-star's own size
# ("≥100 concepts' closed-form edit of all SD-1.4 cross-attn layers", BASELINE.json north_star)
CONFIG_LEGS = ("sd14_erase2p3", "sd14_erase100", "sd14_erase1000p500", "sdxl_debias36x2")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def make_inputs(name: str, device):
    from uce_amd import synth as O
    n_e, n_p, d, table, _ = WORKLOADS[name]
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
    return dict(C=C, G=G, s=s, W=W, mods=mods, rows=rows, d=d, n_e=n_e, n_p=n_p, emb=emb, name=name)


def _log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores() -> int:
    try:
        import psutil
        return int(psutil.cpu_count(logical=False) or 0)
    except Exception:  # noqa: BLE001
        return 0


def cpu_baseline(inp, repeats: int = 5, sweep_threads: bool = True, modules: int = 0, threads: int = 0):
    """The oracle (the reference's own op order on torch CPU: sequential rank-1 fp32 updates, torch.inverse and a
    GEMM per module, uce_sd_erase.py:56-82) timed on the host cores: WHOLE edits of all modules, a fixed thread
    count, the median of `repeats` runs after one untimed run (bounded sample: ~1.5 s per whole 50-concept edit).
    `modules` > 0: a bounded sample for the large configs - only the first `modules` modules of the slab are edited (the
    reference's loop over modules is a plain sequence of independent iterations, uce_sd_erase.py:56) and the time is scaled
    by rows_total / rows_sampled; the line says so.  `threads` > 0: no sweep, that thread count."""
    from oracle import uce_oracle as O
    Wc = inp["W"].cpu()
    ws, off = [], 0
    for _, o in inp["mods"]:
        ws.append(Wc[off:off + o])
        off += o
    n_mod = len(ws)
    scale_up = 1.0
    if modules and modules < n_mod:
        # evenly strided through the table, so that the sample mixes the module widths as the table does (the per-module cost
        # depends on the width: SD-1.4's 32 modules are 320 / 640 / 1280 rows)
        ws = ws[::n_mod // modules][:modules]
        scale_up = n_mod / float(len(ws))
    C, G = inp["C"].cpu(), inp["G"].cpu()
    n_e, n_p = inp["n_e"], inp["n_p"]
    edit = [C[i:i + 1] for i in range(n_e)]
    guide = [G[i:i + 1] for i in range(n_e)]
    pres = [C[n_e + i:n_e + i + 1] for i in range(n_p)]
    ncpu = os.cpu_count() or 1

    def whole():
        t = time.perf_counter()
        O.uce_edit_ref(ws, edit, guide, pres, 1.0, 1.0, 0.5)
        return (time.perf_counter() - t) * scale_up

    # thread sweep, one whole edit each after one untimed run (the reference's op order is ~10^4 small torch ops per edit:
    # it does not scale with cores; SURVEY 8d asks for the count to be stated): the fastest count runs the timed sample.
    # 1 and 8 are in the sweep on purpose: one core is the scalar port, 8 is the survey container's count.
    first = threads if threads > 0 else min(CPU_BASELINE_THREADS, ncpu)
    torch.set_num_threads(first)
    whole()
    sweep = {}
    if threads <= 0 and sweep_threads and repeats >= 5:
        for th in sorted({1, min(CPU_BASELINE_THREADS, ncpu), min(8, ncpu), min(16, ncpu), min(64, ncpu), ncpu}):
            torch.set_num_threads(th)
            sweep[str(th)] = round(whole(), 4)
            if th >= 16 and sweep[str(th)] > 3.0 * min(sweep.values()):
                break                                        # far slower already: do not spend the budget on larger counts
        threads = int(min(sweep, key=sweep.get))
    elif threads <= 0:
        threads = first
    torch.set_num_threads(threads)
    times = []
    t0 = time.perf_counter()
    for _ in range(repeats):
        times.append(whole())
        if time.perf_counter() - t0 > 30.0 and len(times) >= 3:
            break
    med = statistics.median(times)
    n = n_e + n_p
    sample = (f"{len(times)} whole edits (all {n_mod} modules, {n} concepts) after one untimed run" if scale_up == 1.0 else
              f"{len(times)} edits of {len(ws)} of the {n_mod} modules (evenly strided; {n} concepts) after one untimed run, time x {scale_up:.1f} "
              f"(the reference edits module after module)")
    return dict(value=round(n / med, 2), unit="concepts/s", cores=threads, kind="port",
                cpu=cpu_model(), physical_cores=physical_cores(), logical_cpus=ncpu,
                seconds_per_edit=dict(median=round(med, 4), min=round(min(times), 4), max=round(max(times), 4)),
                thread_sweep_seconds_per_edit=sweep,
                sample=f"{sample}, median {med:.3f} s, on {threads} torch threads of {ncpu} logical CPUs")


def cpu_baseline_configs(device, threads: int):
    """The same CPU port for the other single-GPU BASELINE configs, each a bounded sample (a few seconds of host work) at the
    thread count the headline sweep found fastest."""
    out = []
    for name, mods, reps in (("sd14_erase2p3", 0, 3), ("sd14_erase100", 8, 3), ("sd14_erase1000p500", 2, 1), ("sdxl_debias36x2", 4, 1)):
        try:
            b = cpu_baseline(make_inputs(name, "cpu"), repeats=reps, sweep_threads=False, modules=mods, threads=threads)
            out.append({"workload": name, "value": b["value"], "unit": b["unit"], "cores": b["cores"],
                        "seconds_per_edit": b["seconds_per_edit"]["median"], "sample": b["sample"]})
        except Exception as err:  # noqa: BLE001
            out.append({"workload": name, "error": repr(err)})
    return out


# ------------------------------------------------------------------------------------------------------------
# the edit: timed region, per-kernel breakdown, roofline
# ------------------------------------------------------------------------------------------------------------

def edit_path(N: int, n_e: int, d: int, rows: int, algo: int) -> str:
    """Which kernel chain uce_edit takes (mirrors csrc/uce_api.hip:uce_edit)."""
    dual = (algo == 2) or (algo == 0 and round_up(N, 64) < d)
    if not dual:
        return "primal"
    if 1 <= n_e <= N <= 128 and d == 768 and rows >= 1024 and os.environ.get("UCE_EDIT_RESIDENT", "1") != "0":
        return "dual_resident"                                            # one launch, W_old register-resident (uce_edit_resident.hip)
    if 1 <= n_e <= 128 and d in (768, 1024, 2048) and rows >= 1024:      # (UCE_SPLIT_MAX_NE: beyond, Delta + the dense apply)
        return "dual_lowrank"
    return "dual_other"


def step_floor(path: str, N: int, n_e: int, d: int, rows: int):
    """Algorithmic work of one step in its minimal formulation -> (bytes, f32 flop, f64 flop, floor ms)."""
    q = 4.0 * (N + n_e) * d + 8.0 * rows * d
    if path == "primal":
        f32 = 2.0 * rows * d * d
        f64 = (N + 2.0 * n_e) * d * d + (7.0 / 3.0) * d ** 3
    else:
        n = round_up(N, 64)
        f32 = 4.0 * rows * d * n_e
        f64 = 1.0 * N * N * d + n ** 3 / 3.0 + 2.0 * n * n * d
    floor_s = max(q / (HBM_PEAK_GBS * 1e9), f32 / (F32_MFMA_PEAK_TF * 1e12) + f64 / (F64_MFMA_PEAK_TF * 1e12))
    return q, f32, f64, floor_s * 1e3


def kernel_model(name: str, path: str, N: int, n_e: int, d: int, rows: int):
    """(bound, algorithmic work per launch, peak, unit, note) of one kernel of the step."""
    nep, n_sys = round_up(max(n_e, 1), 64), (d if path == "primal" else round_up(N, 64))
    if name == "k_lr_project":
        return "mfma", 2.0 * rows * d * n_e, F32_MFMA_PEAK_TF, "TFLOP/s", (f"f32 MFMA; issues 2*rows*d*{nep} flop on the 64-padded concept tiles; "
                                                                           f"one pass over W per {'64' if nep <= 64 else '128'} concepts")
    if name in ("k_lr_update_s", "k_lr_update"):
        return "hbm", 8.0 * rows * d + 4.0 * rows * nep + 4.0 * n_e * d, HBM_PEAK_GBS, "GB/s", "W in + W out, T in, R once"
    if name == "k_lr_resident":
        # ONE launch = the whole step; W_old is read from HBM once, held in registers, written once.  Both products run on the f16
        # matrix cores with two-term split operands (3 MFMAs per fp32-equivalent product): 3 * 4 * rows * d * 64 flop = 5.9 us of
        # the 2.5 PF/s pipe at SD-1.4's size against 19.2 us of HBM time - the launch is priced against HBM
        byts = 8.0 * rows * d + 4.0 * (N + 2.0 * n_e) * d
        return "hbm", byts, HBM_PEAK_GBS, "GB/s", ("the whole step in one launch (uce_edit_resident.hip): W in once + W out once + the embeddings; W and T stay in "
                                                   f"registers between the two products, which run as {3 * 4.0 * rows * d * nep / 1e9:.1f} GF of f16 MFMA (two-term split "
                                                   "operands, fp32 accumulate); the Gram -> Cholesky -> solve chain rides in rider workgroups and the final "
                                                   "stores cannot start before it ends")
    if name == "k_lr_fused":
        # ONE launch = the whole step: projection + rider chain + update.  Priced against whichever floor is higher
        byts, fl = 8.0 * rows * d + 4.0 * (N + 2.0 * n_e) * d, 4.0 * rows * d * n_e
        if fl / (F32_MFMA_PEAK_TF * 1e12) > byts / (HBM_PEAK_GBS * 1e9):
            return "mfma", fl, F32_MFMA_PEAK_TF, "TFLOP/s", (f"the whole step in one launch: projection + update = 4*rows*d*N_e exact-f32 MFMA flop (the binding floor "
                                                             f"from ~45 concepts on); HBM floor {byts / 8e6:.1f} us for {byts / 1e6:.1f} MB (W in once from HBM, once "
                                                             "from the last-level cache, W out); the Gram -> Cholesky -> solve chain rides in rider blocks")
        return "hbm", byts, HBM_PEAK_GBS, "GB/s", ("the whole step in one launch: W in + W out + the embeddings; T stays in LDS, the second read of W comes "
                                                   f"from the last-level cache; {fl / 1e9:.2f} GF of exact-f32 MFMA ride along")
    if name == "k_trisolve":
        return "mfma", 2.0 * n_sys * n_sys * d, F64_MFMA_PEAK_TF, "TFLOP/s", "f64 MFMA, forward + backward substitution"
    if name == "potrf":
        return "mfma", n_sys ** 3 / 3.0, F64_MFMA_PEAK_TF, "TFLOP/s", ("f64 blocked Cholesky (latency chain of n pivots); on the primal path its "
                                                                       "rider workgroups also compute Bt and the f16 split of W_old")
    if name == "k_gram_primal":
        nb = d // 64                            # (mirrors uce_solve.hip:potrf_la_has_room: 250 - (1 + nb (nb - 1)) >= 64 riders)
        if 3 <= nb <= 14 and n_e > 0:           # the persistent Cholesky launch follows: its rider workgroups compute Bt (and split W_old)
            return "mfma", 1.0 * N * d * d, F64_MFMA_PEAK_TF, "TFLOP/s", ("f64 MFMA, lower tiles of A split over the concepts + slab reduction; "
                                                                          "Bt (2*N_e*d^2 flop) rides in the Cholesky launch")
        return "mfma", (N + 2.0 * n_e) * d * d, F64_MFMA_PEAK_TF, "TFLOP/s", "f64 MFMA, lower tiles of A + all of Bt"
    if name == "k_gram_dual":
        return "mfma", 1.0 * N * N * d, F64_MFMA_PEAK_TF, "TFLOP/s", "f64 MFMA, lower tiles"
    if name == "k_apply_b3":
        return "mfma", 6.0 * 2.0 * rows * d * d, BF16_MFMA_PEAK_TF, "TFLOP/s", "bf16 MFMA, six partial products per fp32-equivalent product (2*rows*d^2 fp32-equivalent flop)"
    if name == "k_apply_h2":
        return "mfma", 3.0 * 2.0 * rows * d * d, BF16_MFMA_PEAK_TF, "TFLOP/s", ("f16 MFMA, three partial products per fp32-equivalent product "
                                                                               "(2*rows*d^2 fp32-equivalent flop); direct-to-LDS operands")
    if name == "k_split_h2":
        return "hbm", 8.0 * rows * d, HBM_PEAK_GBS, "GB/s", "W_old f32 in, two f16 planes out (rides in the Cholesky launch when that is the persistent kernel)"
    if name == "k_split_h2d":
        return "hbm", 8.0 * d * d, HBM_PEAK_GBS, "GB/s", "(I + Delta)^T f32 in, two f16 planes out"
    if name == "k_apply":
        return "mfma", 2.0 * rows * d * d, F32_MFMA_PEAK_TF, "TFLOP/s", "f32 MFMA"
    if name == "k_split3":
        return "hbm", 10.0 * d * d, HBM_PEAK_GBS, "GB/s", "Delta^T f32 in, three bf16 planes out"
    if name == "k_delta_factors":
        return "mfma", 2.0 * n_e * d * d, F32_MFMA_PEAK_TF, "TFLOP/s", "f32 MFMA"
    if name == "k_apply_lowrank_generic":
        return "hbm", 8.0 * rows * d + 8.0 * n_e * d, HBM_PEAK_GBS, "GB/s", "W in + W out"
    return "hbm", 0.0, HBM_PEAK_GBS, "GB/s", "unmodelled"


def _source_hash():
    try:
        from uce_amd import build
        return build.source_hash()
    except Exception:  # noqa: BLE001
        return None


SOURCE_HASH = _source_hash()       # state of csrc/ this run was built from (profiles/traffic.json entries carry theirs)


def load_traffic():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path))
    except Exception:  # noqa: BLE001
        return {}


def kernel_breakdown(H, inp, algo: int, step, iters: int = 30):
    """Per-kernel average duration of the launches of one step (HIP events on the launch stream around every launch,
    recorded by the library itself: uce_profile_begin/_end), each priced against its own roofline."""
    N, d, rows, n_e = inp["C"].shape[0], inp["d"], inp["rows"], inp["n_e"]
    path = edit_path(N, n_e, d, rows, algo)
    prof = H.profile(step, iters)
    traffic = load_traffic().get(inp["name"], {})
    total = sum(ms * cnt for ms, cnt in prof.values()) or 1.0
    rows_out = []
    for name, (ms, cnt) in prof.items():
        kname = "k_lr_update_s" if (name == "k_lr_update" and n_e <= 128) else name
        bound, work, peak, unit, note = kernel_model(kname, path, N, n_e, d, rows)
        ach = work / (ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12) if ms > 0 else 0.0
        ent = dict(kernel=kname, launches_per_step=cnt, avg_ms=round(ms, 5), share=round(ms * cnt / total, 4), bound=bound,
                   achieved=round(ach, 2), peak=peak, unit=unit, frac=round(ach / peak, 4),
                   **({"algorithmic_bytes": work} if bound == "hbm" else {"algorithmic_flops": work}), note=note)
        t = traffic.get(kname)
        ent["traffic"] = t.get("chain_bytes", t.get("total_bytes")) if isinstance(t, dict) else None
        if isinstance(t, dict) and t.get("mfma_util") is not None:
            ent["mfma_util"] = t["mfma_util"]
        if isinstance(t, dict):
            # the counters are folded from separate rocprofv3 passes (profiles/traffic.json): say which library they saw
            ent["traffic_src"] = t.get("src")
            ent["traffic_stale"] = bool(t.get("src") is not None and t.get("src") != SOURCE_HASH)
        rows_out.append(ent)
    rows_out.sort(key=lambda e: -e["share"])
    if path == "dual_lowrank" and N <= 128:
        # inside uce_edit the projection launch also carries the whole small-system chain (Gram -> Cholesky ->
        # triangular solves in rider blocks) and ends when the LONGER of the two ends: time the GEMM on its own too
        Dm = (inp["G"] - inp["C"][:n_e]).contiguous()
        ms = time_kernel(lambda: H.lowrank_project(inp["W"], Dm), iters)
        for e in rows_out:
            if e["kernel"] == "k_lr_project":
                fl = e["algorithmic_flops"]
                e["gemm_alone"] = dict(avg_ms=round(ms, 5), achieved=round(fl / (ms * 1e-3) / 1e12, 2),
                                       frac=round(fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4),
                                       frac_issued=round(2.0 * rows * d * round_up(n_e, 64) / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4))
                e["note"] += ("; in uce_edit this launch also carries the Gram + Cholesky + triangular-solve rider blocks and "
                              "lasts as long as that latency chain (tools/dbg_chain.py); gemm_alone = the projection launched by itself")
    return path, rows_out, total


def time_steps(step, steps: int, world: int, device):
    """EXACTLY `steps` steps between barrier + synchronize on both sides (wall clock, max over ranks) and, inside the
    same region, HIP events on the launch stream (sub-100-us steps: the events exclude the host-side sync latency)."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the first timing event of a process costs ~40 ms of one-off runtime set-up: spend it outside the timed region
    # (seen as 0.28 ms/step "wall" against 0.064 ms/step of events when the edit leg was the first thing a run did)
    warm = torch.cuda.Event(enable_timing=True)
    warm.record()
    warm.synchronize()
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    ev = e0.elapsed_time(e1) * 1e-3
    if world > 1:
        t = torch.tensor([elapsed, ev], dtype=torch.float64, device=_scalar_device(device))
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, ev = float(t[0].item()), float(t[1].item())
    return elapsed, ev


def run_edit(H, name: str, device, steps: int, warmup: int, algo: int, world: int = 1, breakdown: bool = True):
    inp = make_inputs(name, device)
    C, G, s, W = inp["C"], inp["G"], inp["s"], inp["W"]
    N, d, rows, n_e = C.shape[0], inp["d"], inp["rows"], inp["n_e"]
    out = torch.empty_like(W)
    H.reserve(d, max(N, d))
    H.reserve_rows(rows, max(n_e, 1))

    def step():
        H.edit(C, G, s, 0.5, W, out=out, algo=algo)

    _log(f"{name}: inputs ready (N={N} d={d} rows={rows}); warm-up")
    for _ in range(warmup):
        step()
    H.status()
    elapsed, ev = time_steps(step, steps, world, device)
    H.status()
    ms = 1e3 * elapsed / steps
    _log(f"{name}: {ms:.4f} ms/step (wall), {1e3 * ev / steps:.4f} ms/step (events)")
    res = dict(inp=inp, out=out, elapsed=elapsed, ms_per_step=ms, ms_per_step_events=1e3 * ev / steps, N=N)
    if breakdown:
        path, kernels, _ = kernel_breakdown(H, inp, algo, step)
        q, f32, f64, floor_ms = step_floor(path, N, n_e, d, rows)
        tr = [k["traffic"] for k in kernels]
        res.update(path=path, kernels=kernels, floor_ms=floor_ms, alg_bytes=q, alg_f32=f32, alg_f64=f64,
                   traffic=sum(tr) if tr and all(t is not None for t in tr) else None)
    return res


def roofline_block(r):
    """`roofline` of the JSON line: the kernel with the largest time share on top, the rest under `kernels`."""
    top = dict(r["kernels"][0])
    step_ms = r["ms_per_step_events"]
    top.update(kernels=r["kernels"][1:], path=r["path"],
               step_floor_ms=round(r["floor_ms"], 5), step_frac=round(r["floor_ms"] / step_ms, 4),
               step_algorithmic_bytes=r["alg_bytes"], step_algorithmic_flops={"f32": r["alg_f32"], "f64": r["alg_f64"]},
               step_traffic=r["traffic"],
               step_traffic_ratio=round(r["traffic"] / r["alg_bytes"], 3) if r["traffic"] else None,
               traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950), bytes per "
                              "launch (profiles/traffic.json <- tools/pmc_fold.py); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                              "(32 * SQ_BUSY_CYCLES)")
    return top


def r_small(name: str) -> bool:
    return WORKLOADS[name][2] == 768 and WORKLOADS[name][0] + WORKLOADS[name][1] <= 128


def config_leg(H, name: str, device, algo: int):
    n_e, n_p, d, _, idx = WORKLOADS[name]
    # enough steps that the timed region is tens of milliseconds (20 steps of a 0.1 ms edit measured 10-15 % slow: the first
    # launches after a workload switch run on cold caches and a ramping clock)
    r = run_edit(H, name, device, 200 if r_small(name) else 100, 20, algo)
    k0 = r["kernels"][0]
    out = dict(baseline_config=idx, workload=name, concepts=n_e + n_p, d=d, rows=r["inp"]["rows"],
               ms_per_step=round(r["ms_per_step"], 5), ms_per_step_events=round(r["ms_per_step_events"], 5),
               concepts_per_s=round((n_e + n_p) / (r["ms_per_step_events"] * 1e-3), 1), path=r["path"],
               step_floor_ms=round(r["floor_ms"], 5), step_frac=round(r["floor_ms"] / r["ms_per_step_events"], 4),
               step_algorithmic_bytes=r["alg_bytes"], step_traffic=r["traffic"],
               step_traffic_ratio=round(r["traffic"] / r["alg_bytes"], 3) if r["traffic"] else None,
               dominant=dict(kernel=k0["kernel"], avg_ms=k0["avg_ms"], share=k0["share"], bound=k0["bound"],
                             achieved=k0["achieved"], peak=k0["peak"], unit=k0["unit"], frac=k0["frac"]),
               kernels=[dict(kernel=k["kernel"], avg_ms=k["avg_ms"], launches_per_step=k["launches_per_step"], share=k["share"],
                             frac=k["frac"], bound=k["bound"], traffic=k["traffic"], **({"mfma_util": k["mfma_util"]} if "mfma_util" in k else {}))
                        for k in r["kernels"]])
    # the arithmetic type of the leg's products (the headline's `dtype` covers the project + update form of the small configs)
    names = " ".join(k["kernel"] for k in out["kernels"])
    if "apply_h2" in names or "k_apply_h2" in names:
        out["dtype"] = ("f64 Gram / Cholesky / inverse; dense apply W_old (I + Delta) in f32 via a two-term f16 split of both operands "
                        "(2 x 11 significand bits per operand under per-row / per-column power-of-two scales, f32 accumulate; "
                        "3.2e-7 from fp64 on this config)")
    else:
        out["dtype"] = "f32 products on the f32 matrix cores, f64 Gram / Cholesky / solves"
    del r
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------
# generation and attention legs
# ------------------------------------------------------------------------------------------------------------

def _sync(device) -> None:
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def _scalar_device(device):
    """Where the few timing scalars of the collectives live: the device under RCCL, the host under gloo."""
    import torch.distributed as dist
    return torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")


def count_generation_flops(pipe, device, steps: int):
    """Algorithmic FLOPs of ONE 512 x 512 image through the build's own pipeline ("recount from the build's own
    U-Net"): every matrix product the U-Net issues for one prompt's CFG pair (counted by wrapping the UceHandle entry points during
    one call: 2 M N K per linear layer, 2 M 9 Cin Cout per 3x3 convolution, 4 B L^2 C / 4 B Lq Lk C for self- / cross-attention),
    x (steps + 1) U-Net calls of the PNDM schedule, + one VAE decode.  Norms, activations and the text encoder are not counted (the
    figure is the MFMA work the roofline fraction is quoted against)."""
    from uce_amd import edit as E
    fam = {"conv": 0.0, "linear": 0.0, "sattn": 0.0, "xattn": 0.0}
    depth = [0]
    orig = {}

    def prod(shape):
        n = 1
        for d in shape:
            n *= int(d)
        return n

    def account(name, a, k):
        if name in ("linear", "linear_colscale"):
            x, w = a[0], a[1]
            K = w.shape[1]
            fam["linear"] += 2.0 * (prod(x.shape) // x.shape[-1]) * w.shape[0] * K
            if k.get("x2") is not None:
                pass                                                        # (the weight already spans both sources)
        elif name == "linear_f32":
            x, w = a[0], a[1]
            fam["linear"] += 2.0 * x.shape[0] * w.shape[0] * x.shape[1]
        elif name in ("conv3x3_nhwc", "conv3x3_igemm"):
            x, w = a[0], a[1]
            up = bool(k.get("upsample", False))
            st = int(k.get("stride", 1))
            Hh, Ww = (2 * x.shape[2], 2 * x.shape[3]) if up else (x.shape[2] // st, x.shape[3] // st)
            fam["conv"] += 2.0 * x.shape[0] * Hh * Ww * 9 * x.shape[1] * w.shape[0]
        elif name == "conv3x3_c4":
            x, w = a[0], a[1]
            fam["conv"] += 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * 36 * w.shape[0]
        elif name in ("sattn_packed", "sattn_packed_exp2"):
            B, L, C3 = a[0].shape
            fam["sattn"] += 4.0 * B * L * L * (C3 // 3)
        elif name == "sattn":
            q, kk = a[0], a[1]
            fam["sattn"] += 4.0 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]
        elif name == "xattn":
            q, kk = a[0], a[1]
            fam["xattn"] += 4.0 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]

    def wrap(name):
        f = getattr(E.UceHandle, name)
        orig[name] = f

        def g(self, *a, **k):
            if depth[0] == 0:
                account(name, a, k)
            depth[0] += 1
            try:
                return f(self, *a, **k)
            finally:
                depth[0] -= 1
        setattr(E.UceHandle, name, g)

    names = ("linear", "linear_f32", "linear_colscale", "conv3x3_nhwc", "conv3x3_igemm", "conv3x3_c4", "sattn_packed", "sattn_packed_exp2",
             "sattn", "xattn")
    for n in names:
        wrap(n)
    try:
        x = torch.randn(2, 4, 64, 64, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        t = torch.tensor([500.0, 500.0], device=device)
        ctx = torch.randn(2, 77, pipe.unet.cfg.cross_attention_dim, device=device).to(torch.bfloat16)
        saved = [(m, m.kv_cache) for m in pipe.unet.modules() if hasattr(m, "kv_cache")]
        pipe.unet.cache_context(ctx)                                        # hoisted: once per image, not per call
        once = dict(fam)
        for kf in fam:
            fam[kf] = 0.0
        pipe.unet(x, t, ctx)
        for m, c in saved:
            m.kv_cache = c
        unet = dict(fam)
        for kf in fam:
            fam[kf] = 0.0
        vae = {}
        if getattr(pipe, "vae", None) is not None:
            pipe.vae.decode(torch.randn(1, 4, 64, 64, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
            vae = dict(fam)
    finally:
        for n, f in orig.items():
            setattr(E.UceHandle, n, f)
    torch.cuda.synchronize()
    per_call = sum(unet.values())
    per_image = per_call * (steps + 1) + sum(vae.values()) + sum(once.values())
    tot = {k: unet[k] * (steps + 1) + vae.get(k, 0.0) + once.get(k, 0.0) for k in unet}
    return {"unet_call_cfg_pair": per_call, "unet_calls": steps + 1, "vae_decode": sum(vae.values()),
            "context_projections_once": sum(once.values()), "per_image": per_image, "per_image_by_family": tot}


def generation_roofline(pipe, device, steps: int, seconds_per_image: float, batch: int):
    """The images/s half of the metric against the chip: counted MFMA FLOPs per image (count_generation_flops) over the measured
    seconds per image of ONE rank -> achieved PFLOP/s and its fraction of the dense bf16 peak; per kernel family, the counted FLOPs
    over the family's share of the image time (shares from the committed steady-state rocprofv3 profile of the same loop at the
    same prompts per call, profiles/r06/generate_families_b<batch>.json - else the round-5 one - when there is one)."""
    fl = count_generation_flops(pipe, device, steps)
    per_image = fl["per_image"]
    ach = per_image / seconds_per_image / 1e15
    out = {"bound": "mfma", "unit": "PFLOP/s", "algorithmic_flops_per_image": per_image,
           "flops_per_unet_call_cfg_pair": fl["unet_call_cfg_pair"], "unet_calls_per_image": fl["unet_calls"],
           "vae_decode_flops": fl["vae_decode"], "flops_per_image_by_family": fl["per_image_by_family"],
           "seconds_per_image": round(seconds_per_image, 5), "achieved": round(ach, 4), "peak": BF16_MFMA_PEAK_TF / 1e3,
           "frac": round(ach / (BF16_MFMA_PEAK_TF / 1e3), 4),
           "floor_seconds_per_image": round(per_image / (BF16_MFMA_PEAK_TF * 1e12), 5),
           "note": "FLOPs counted from the build's own U-Net / VAE (2 M N K per linear layer and convolution tap, 4 B L Lk C per attention); "
                   "norms, activations, text encoder, host work are in the seconds but not in the FLOPs"}
    for rnd, cand in [(r, c) for c in (batch, 128, 64) for r in ("r06", "r05")]:
        path = os.path.join(ROOT, "profiles", rnd, f"generate_families_b{cand}.json")
        if os.path.exists(path):
            try:
                fam = json.load(open(path))
                shares = {k: v["share"] for k, v in fam["families"].items()}
                out["families"] = {
                    k: {"time_share": shares.get(k), "flops_share": round(fl["per_image_by_family"].get(k, 0.0) / per_image, 4),
                        "achieved_PFs": (round(fl["per_image_by_family"][k] / (shares[k] * seconds_per_image) / 1e15, 4)
                                         if k in fl["per_image_by_family"] and shares.get(k) else None)}
                    for k in sorted(set(shares) | set(fl["per_image_by_family"]))}
                out["families_profile"] = os.path.relpath(path, ROOT)
                out["families_profile_prompts_per_call"] = cand
                out["families_note"] = ("time shares: kernel time of the steady denoising loop under rocprofv3 (eager launches); per-family PF/s = "
                                        "counted FLOPs / (share x seconds per image), i.e. it charges the family with its share of the "
                                        "host / VAE time too")
            except Exception as err:  # noqa: BLE001
                out["families"] = {"error": repr(err)}
            break
    return out


def generation_leg(device, world, n_images, steps, edited_slab, batch=8, model_id="CompVis/stable-diffusion-v1-4",
                   dtype=torch.bfloat16, vae=True, rowwise_images=0, keep_pipe=None):
    """Secondary figure (BASELINE.json's second metric): images/s of the edited SD-1.4 pipeline,
    512x512, `steps` PNDM steps (+1 U-Net call), guidance 7.5, bf16, CPU-seeded latents, synthetic
    (seeded-random) weights, cross-attention through uce_xattn_fwd.  Every rank generates its own
    `n_images` prompts (the sharded rows of evalscripts/generate-images-sd.py); the edited attn2
    weights are broadcast from rank 0 first when world > 1.

    Collective sequence (every rank passes every one of them, whatever happens on it - a rank that raised would
    otherwise leave the others waiting until the RCCL time-out):  broadcast(weights) | barrier | timed loop | barrier |
    all_reduce(MAX: seconds, failure flag) | all_gather(per-rank seconds).  UCE_BENCH_FAIL_RANK=r makes rank r raise
    inside the timed loop (tests/test_generate_cpu.py drives this with gloo, world 2)."""
    from uce_amd.sd import pipeline as sdp
    from uce_amd import edit as E
    _log("generation leg: building the synthetic pipeline")
    pipe = sdp.load_pipeline(model_id, dtype, device, synthetic=True, vae=vae)
    rank = int(os.environ.get("RANK", "0"))
    fail_rank = int(os.environ.get("UCE_BENCH_FAIL_RANK", "-1"))
    bcast_ms = None
    failure = None
    if edited_slab is not None:
        blob = edited_slab.clone()
        if world > 1:
            _sync(device)
            tb = time.perf_counter()
            torch.distributed.broadcast(blob, src=0)          # RCCL over xGMI: the one exchange step
            _sync(device)
            bcast_ms = 1e3 * (time.perf_counter() - tb)
        try:
            mods = E.collect_uce_modules(pipe.unet)
            off = 0
            state = {}
            for n, m in mods:
                r = m.weight.shape[0]
                state[n + ".weight"] = blob[off:off + r]
                off += r
            sdp.patch_unet(pipe, state)
        except Exception as err:  # noqa: BLE001
            failure = f"weight patch: {err!r}"
    batch = max(1, min(batch, n_images))

    # the rows this rank would take of the prompt table: real records of the reference's data/coco_30k.csv first (the
    # committed fixture tests/golden/coco30k_rows.csv, with their own evaluation seeds), then rows of the same schema
    # from the caption grammar (tools/make_prompts_csv.py); row r of every `world` belongs to rank r
    table = prompt_table(world * max(n_images, rowwise_images))
    mine = table[rank::world]
    extra = {} if vae else {"output_type": "latent"}

    def run(first, count, nsteps):
        prompts = [mine[first + j][2] for j in range(count)]
        gens = [torch.Generator().manual_seed(mine[first + j][3]) for j in range(count)]   # evaluation_seed, CPU generator
        return pipe(prompts if count > 1 else prompts[0], num_inference_steps=nsteps, guidance_scale=7.5,
                    generator=gens if count > 1 else gens[0], **extra)

    chunks = [(lo, min(batch, n_images - lo)) for lo in range(0, n_images, batch)]
    if failure is None:
        try:
            for count in sorted({c for _, c in chunks}):          # warm-up: solver search, hipGraph capture
                run(0, count, 2)
        except Exception as err:  # noqa: BLE001
            failure = f"warm-up: {err!r}"
    # The reference's loop ends in `im.save(...)` (generate-images-sd.py:45-46): the PNGs are encoded and written INSIDE the timed
    # region, on worker threads behind the next batch's denoising exactly as uce_amd.generate.generate_images does it (same default
    # worker count, capped so that `world` ranks never ask for more threads than the node has cores), and the clock stops only
    # when the last file is on disk.
    from uce_amd import generate as _gen
    png_dir = tempfile.mkdtemp(prefix="uce_bench_png_") if vae else None
    # (after the rank was pinned to its block of cores; UCE_BENCH_PNG_WORKERS: a fixed count for an A/B)
    png_workers = max(1, _gen.png_worker_count(int(os.environ.get("UCE_BENCH_PNG_WORKERS", _gen.PNG_WORKERS_AUTO)), world))
    writer = ThreadPoolExecutor(max_workers=png_workers) if vae else None
    pending = []
    png_cpu_s = [0.0]

    def save_png(im, path):
        t = time.thread_time()
        im.save(path)
        png_cpu_s[0] += time.thread_time() - t                       # (a float add under the GIL: good enough for a report)

    if world > 1:
        torch.distributed.barrier()
    _sync(device)
    cpu0 = os.times()
    t0 = time.perf_counter()
    if failure is None:
        try:
            for lo, count in chunks:
                if rank == fail_rank:
                    raise RuntimeError("injected failure (UCE_BENCH_FAIL_RANK)")
                res = run(lo, count, steps)                       # -> PIL images on the host, as pipe(...).images
                if writer is not None:
                    for j, im in enumerate(res.images):
                        pending.append(writer.submit(save_png, im, os.path.join(png_dir, f"{mine[lo + j][0]}_0.png")))
            for f in pending:
                f.result()
        except Exception as err:  # noqa: BLE001
            failure = f"timed loop: {err!r}"
    _sync(device)
    mine_s = time.perf_counter() - t0                             # this rank's own loop, before it waits for the others
    if writer is not None:
        writer.shutdown(wait=True)
    png_bytes = 0
    if png_dir is not None:
        try:
            png_bytes = sum(os.path.getsize(os.path.join(png_dir, f)) for f in os.listdir(png_dir))
        finally:
            shutil.rmtree(png_dir, ignore_errors=True)
    cpu1 = os.times()
    host_cpu_s = (cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)   # this process, all its threads (CPU RNG, tokenizer, image conversion)
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0                                 # the job's timed region ends HERE (the row-wise figure below is its own)
    # the drop-in CLI's default is ONE prompt per pipe() call (generate-images-sd.py:29-42): the same loop row by row, on a
    # few rows (its own captured step at batch 2)
    rowwise = None
    if failure is None and rowwise_images > 0:
        try:
            run(0, 1, 2)
            _sync(device)
            tr = time.perf_counter()
            for j in range(rowwise_images):
                run(j, 1, steps)
            _sync(device)
            rowwise = rowwise_images / (time.perf_counter() - tr)
        except Exception as err:  # noqa: BLE001
            _log(f"row-wise leg failed: {err!r}")
    per_rank = [mine_s]
    if world > 1:
        t = torch.tensor([el, 0.0 if failure is None else 1.0], dtype=torch.float64, device=_scalar_device(device))
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t[0].item())
        if failure is None and t[1].item() > 0:
            failure = "another rank failed"
        mine_t = torch.tensor([mine_s], dtype=torch.float64, device=_scalar_device(device))
        gathered = [torch.zeros_like(mine_t) for _ in range(world)]
        torch.distributed.all_gather(gathered, mine_t)
        per_rank = [float(g.item()) for g in gathered]
    if failure is not None:
        _log("generation leg failed: " + failure)
        return {"metric": "images/sec 512x512 50-step", "value": None, "error": failure, "n_gpus": world}
    out = {"metric": "images/sec 512x512 50-step", "value": round(world * n_images / el, 4), "unit": "images/s",
           "n_gpus": world, "images_per_rank": n_images, "prompts_per_unet_call": batch, "steps": steps,
           "dtype": str(dtype).replace("torch.", ""), "scaling": "weak",
           "data": "synthetic weights; prompts = real coco_30k.csv records (tests/golden/coco30k_rows.csv, their own seeds) then "
                   "same-schema synthetic captions; CPU-seeded latents",
           "seconds": round(el, 3),
           "per_rank_images_per_s": [round(n_images / s, 4) for s in per_rank],
           # host side of one rank: CPU seconds (user + system, every thread of the process) per image inside the timed loop -
           # the CPU-seeded latent draws, tokenisation, launch / graph-replay calls and the uint8 / PIL conversion; an 8-rank
           # node needs 8 x this per wall second of generation from its host cores
           "host_cpu_seconds_per_image": round(host_cpu_s / max(n_images, 1), 4),
           "host_cpu_cores_busy": round(host_cpu_s / max(mine_s, 1e-9), 3)}
    if vae:
        out["png"] = {"in_timed_region": True, "workers_per_rank": png_workers, "cpu_seconds_per_image": round(png_cpu_s[0] / max(n_images, 1), 4),
                      "bytes_per_image": int(png_bytes / max(n_images, 1)),
                      "note": "im.save of every image (generate-images-sd.py:45-46) on worker threads; the clock stops after the last file"}
    if rowwise is not None:
        out["rowwise"] = {"value": round(world * rowwise, 4), "unit": "images/s", "prompts_per_unet_call": 1, "images": rowwise_images,
                          "note": "--batch_prompts 1: one pipe() call per CSV row like generate-images-sd.py:29-42 (every layer on the "
                                  "few-tile kernel forms; no library GEMM / convolution). The CLI default (--batch_prompts 0) batches rows "
                                  "automatically and runs at `value` above" + (f"; rank 0's figure x {world}" if world > 1 else "")}
    if rank == 0:
        try:
            out["roofline"] = generation_roofline(pipe, device, steps, el / max(world * n_images, 1) * world, batch)
        except Exception as err:  # noqa: BLE001
            out["roofline"] = {"error": repr(err)}
    if bcast_ms is not None:
        out["weight_broadcast_ms"] = round(bcast_ms, 3)
    if keep_pipe is not None:
        keep_pipe.append(pipe)
    return out


def uce_wall_leg(pipe_bf16, device, tmpdir: str):
    """The number the reference itself prints (uce_sd_erase.py:90-91 "Model edited in X seconds"; README.md:5 "under 1 second"):
    edit.UCE() END TO END on the synthetic SD-1.4 pipeline - module discovery + slab, one text-encoder forward per unique
    concept string (or --embed_batch strings per forward), the closed-form edit, device -> host + safetensors - for BASELINE
    configs 0-2, stage by stage.

    The pipeline is loaded exactly as the drop-in CLI (and the reference, trainscripts/uce_sd_erase.py:24,117,197-200) loads it:
    **fp32**, no VAE.  The same legs on the bf16 pipeline of the generation leg ride along as `bf16_pipeline` - a labelled
    extra (a user who edits the pipeline they generate with), never the headline of this block."""
    from uce_amd import edit as E
    from uce_amd.sd import pipeline as sdp
    import contextlib
    import io

    def legs(pipe, which, reps=1, per_string=True):
        out = []
        for name, n_e, n_p in which:
            edit = [f"artist number {i}" for i in range(n_e)]
            pres = [f"kept artist {i}" for i in range(n_p)]
            guide = ["art"] * n_e
            ent = {"workload": name, "concepts": n_e + n_p}
            modes = (("default", None, reps),) + ((("per_string", 0, 1),) if per_string else ())   # default = automatic batching
            for label, eb, n in modes:
                runs = []
                for _ in range(n):
                    tm = {}
                    with contextlib.redirect_stdout(io.StringIO()):
                        E.UCE(pipe, edit, guide, pres, 1.0, 1.0, 0.5, tmpdir, f"wall_{name}_{label}", device=str(device), embed_batch=eb,
                              timings=tm)
                    runs.append({k: round(v, 4) for k, v in tm.items()})
                runs.sort(key=lambda t: t["total"])
                ent[label] = runs[len(runs) // 2]                               # the median call
                if n > 1:
                    ent[label + "_totals"] = [t["total"] for t in runs]
            out.append(ent)
        return out

    all_cfg = (("sd14_erase2p3", 2, 3), ("sd14_erase50", 50, 0), ("sd14_erase1000p500", 1000, 500))
    res = {"metric": "UCE() wall seconds, end to end (the reference's own printed figure)", "unit": "s",
           "stages": "slab | embed | edit | save | total"}
    extra = legs(pipe_bf16, all_cfg[1:], per_string=False) if pipe_bf16 is not None else None
    del pipe_bf16
    torch.cuda.empty_cache()
    pipe = sdp.load_pipeline("CompVis/stable-diffusion-v1-4", torch.float32, device, synthetic=True, vae=False)
    legs(pipe, all_cfg[:1], per_string=False)                                # untimed: first-call set-up of the fp32 encoder
    res["text_encoder"] = ("CLIP-L architecture, seeded-random weights, fp32 (the CLI's / reference's load: torch_dtype float32, no VAE), "
                           "on the GPU; batched mode runs it on token positions 0 .. max(last-token index) only (causal encoder: "
                           "edit.last_token_embeddings)")
    first = legs(pipe, all_cfg, per_string=False)   # the first UCE() of each size in this process: includes the GEMM library's one-time set-up per new shape
    res["configs"] = legs(pipe, all_cfg, reps=3)    # ... and the same calls again (median of three: a shared host adds 10-150 ms at will)
    res["first_call_configs"] = first
    res["note"] = ("`configs` = the median of three further UCE() calls of each size in the process (all totals listed), `first_call_configs` = "
                   "the first (torch's GEMM library picks a kernel per new shape on first use: ~50 ms at 1 500 concepts)")
    if extra is not None:
        res["bf16_pipeline"] = {"text_encoder": "the generation leg's bf16 pipeline (NOT what the CLI loads): labelled extra",
                                "configs": extra}
    del pipe
    torch.cuda.empty_cache()
    return res


def prompt_table(n: int):
    """`n` rows (case_number, source, prompt, evaluation_seed, coco_id): the real coco_30k records of the fixture, then
    synthetic rows of the same schema."""
    from uce_amd import synth
    rows = []
    path = os.path.join(ROOT, "tests", "golden", "coco30k_rows.csv")
    if os.path.exists(path):
        import csv
        with open(path, newline="") as fh:
            rd = csv.reader(fh)
            next(rd)
            rows = [(int(r[0]), r[1], r[2], int(r[3]), int(r[4])) for r in rd][:n]
    if len(rows) < n:
        rows += synth.coco_like_rows(n - len(rows), seed=0, first_case=30000)
    return rows


XATTN_SHAPES = ((4096, 40), (1024, 80), (256, 160), (64, 160))
TIME_KERNEL_BURST_MAX = 100     # untimed launches in front of a timed burst (time_kernel); tools/pmc_fold.py reads `launches` off the line


def xattn_algorithmic_bytes(B: int, Lq: int, C: int, Lk: int = 77) -> float:
    """Q in + O out + K, V in, bf16, once each: what one k_xattn launch has to move."""
    return 2.0 * (B * Lq * C * 2) + 2.0 * (B * Lk * C * 2)


def sattn_algorithmic_bytes(B: int, L: int, C: int) -> float:
    """q, k, v in + o out, bf16, once each."""
    return 4.0 * B * L * C * 2


def sattn_algorithmic_flops(B: int, L: int, C: int) -> float:
    return 4.0 * B * L * L * C


def kernel_launches(iters: int, burst=None) -> int:
    """How many times time_kernel(fn, iters, burst) calls fn (untimed burst + timed launches)."""
    b = max(1, min(iters, TIME_KERNEL_BURST_MAX)) if burst is None else int(burst)
    return b + iters


def xattn_leg(device, batches=(2, 16), iters: int = 100):
    """Cross-attention kernel alone at SD-1.4's four attn2 shapes (H = 8, Lk = 77, bf16) at B = 2 (the CFG pair
    of one prompt) and at the batch the generation leg runs (2 x prompts per U-Net call): algorithmic bytes =
    Q + O + K + V once, per launch, vs the HBM peak."""
    from uce_amd import edit as E
    H = E.UceHandle.get(device)
    traffic = load_traffic().get("xattn", {})
    out = []
    for B in batches:
        for Lq, dh in XATTN_SHAPES:
            C = 8 * dh
            q = torch.randn(B, Lq, C, device=device).bfloat16()
            k = torch.randn(B, 77, C, device=device).bfloat16()
            v = torch.randn_like(k)
            o = torch.empty_like(q)
            ms = time_kernel(lambda: H.xattn(q, k, v, 8, out=o), iters)
            byts = xattn_algorithmic_bytes(B, Lq, C)
            ent = {"B": B, "Lq": Lq, "dh": dh, "avg_us": round(ms * 1e3, 2), "bytes": byts, "launches": kernel_launches(iters),
                   "achieved_GBs": round(byts / (ms * 1e-3) / 1e9, 1),
                   "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            t = traffic.get(f"B{B}_Lq{Lq}_dh{dh}")
            if isinstance(t, dict):
                ent["traffic"] = t.get("total_bytes")
                ent["traffic_stale"] = bool(t.get("src") is not None and t.get("src") != SOURCE_HASH)
                if t.get("mfma_util") is not None:
                    ent["mfma_util"] = t["mfma_util"]
            out.append(ent)
    return {"kernel": "k_xattn", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "shapes": out,
            "note": "B = 2 launches move 1.4-10.7 MB each (launch/latency-bound); the batched rows are the ones "
                    "the generation leg issues"}


def sattn_leg(device, B, iters: int = 10, with_torch: bool = True):
    """Self-attention kernel (attn1 of the U-Net, uce_sattn_fwd) at SD-1.4's four shapes and the generation batch,
    beside torch's scaled_dot_product_attention on the same tensors; 4*B*H*L^2*dh flop per call vs the dense bf16
    MFMA peak (the loop is VALU-bound on the online softmax, not MFMA-bound)."""
    import torch.nn.functional as F
    from uce_amd import edit as E
    H = E.UceHandle.get(device)
    traffic = load_traffic().get("sattn", {})
    out = []
    for L, dh in XATTN_SHAPES:
        C = 8 * dh
        q = torch.randn(B, L, C, device=device).bfloat16()
        k, v = torch.randn_like(q), torch.randn_like(q)
        o = torch.empty_like(q)
        ms = time_kernel(lambda: H.sattn(q, k, v, 8, out=o), iters)
        fl = sattn_algorithmic_flops(B, L, C)
        ent = {"B": B, "L": L, "dh": dh, "avg_us": round(ms * 1e3, 1), "launches": kernel_launches(iters),
               "bytes": sattn_algorithmic_bytes(B, L, C), "achieved_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1),
               "frac": round(fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4)}
        if with_torch:
            sp = lambda t: t.view(B, L, 8, dh).transpose(1, 2)  # noqa: E731
            ent["torch_sdpa_us"] = round(time_kernel(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)), iters) * 1e3, 1)
            # peaked logits (q, k x 5: scores 25x wider, the running maximum keeps moving and the rescale branch of the lazy
            # maximum is taken far more often than on exchangeable scores) - the same kernel, the other end of its data dependence
            qp, kp = (q.float() * 5).bfloat16(), (k.float() * 5).bfloat16()
            ent["peaked_logits_us"] = round(time_kernel(lambda: H.sattn(qp, kp, v, 8, out=o), iters) * 1e3, 1)
            ent["peaked_logits_torch_sdpa_us"] = round(
                time_kernel(lambda: F.scaled_dot_product_attention(sp(qp), sp(kp), sp(v)), iters) * 1e3, 1)
            del qp, kp
        ent["unet_dispatch"] = "uce_sattn_packed_fwd"      # every attn1 layer, whatever its length (sd/unet.py: no library attention)
        if with_torch and H.sattn_exp2_form(B, 8, L, dh):    # (not in the --only sattn passes: tools/pmc_fold.py counts their launches)
            # what the U-Net issues for this layer: the packed projection writes q * dh^-0.5 * log2(e) (uce_linear_colscale_fwd, one
            # rounding) and the attention kernel takes its scores as exp2 arguments.  Timed on the same values (q scaled in f32, then
            # rounded); tests/test_sattn_gpu.py holds its parity against fp64.
            c = dh ** -0.5 * 1.4426950408889634
            qkv = torch.cat([(q.float() * c).bfloat16(), k, v], dim=-1)
            ms2 = time_kernel(lambda: H.sattn_packed_exp2(qkv, 8), iters)
            ent["exp2_domain_us"] = round(ms2 * 1e3, 1)
            ent["exp2_domain_frac"] = round(fl / (ms2 * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4)
            ent["unet_dispatch"] = "uce_linear_colscale_fwd + uce_sattn_packed_exp2_fwd"
            del qkv
        t = traffic.get(f"B{B}_L{L}_dh{dh}")
        if isinstance(t, dict):
            ent["traffic"] = t.get("total_bytes")
            ent["traffic_stale"] = bool(t.get("src") is not None and t.get("src") != SOURCE_HASH)
            if t.get("mfma_util") is not None:
                ent["mfma_util"] = t["mfma_util"]
        out.append(ent)
    return {"kernel": "k_sattn (+ k_vt)", "bound": "mfma", "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "shapes": out}


def time_kernel(fn, iters: int, burst=None):
    """Average duration (ms) of `fn`'s launches on the current stream, HIP events around `iters`
    back-to-back launches (the library enqueues on torch's current stream).  The launches of an untimed warm-up burst come first
    (as many as are timed, at most 100): after an idle gap the first handful of launches of an HBM-bound kernel run ~15 % faster
    than the hundredth (k_xattn_g<40> at B = 128: 137 us for launches 2-5, 165 us median over the first hundred, 145-150 us from
    there on while the shader clock settles from 2.40 to ~2.15 GHz - profiles/r05/xattn_timing_bursts.json; rocprofv3's four-launch
    passes see the first figure), the steady state is what a generation loop runs at."""
    for _ in range(kernel_launches(iters, burst) - iters):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# ------------------------------------------------------------------------------------------------------------

def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run
    (one per GPU, rendezvous on 127.0.0.1) and pass their output through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    _log("self-launch: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="sd14_erase50", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", default="auto", choices=["auto", "primal", "dual"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-images", type=int, default=256,
                    help="images per rank for the secondary images/s figure (0 = skip)")
    ap.add_argument("--gen-batch", type=int, default=128, help="prompts denoised per U-Net call (measured on one MI355X with the final "
                    "kernels of round 4: 64 -> 9.30, 96 -> 9.32, 128 -> 9.54 images/s; round 3: 16 -> 6.95, 32 -> 7.75; the CLI keeps the "
                    "reference's row-by-row default, reported as `rowwise`)")
    ap.add_argument("--gen-rowwise", type=int, default=4, help="images of the row-by-row (one prompt per call) figure; 0 = skip")
    ap.add_argument("--gen-steps", type=int, default=50)
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configs")
    ap.add_argument("--only", default="", choices=["", "edit", "xattn", "sattn", "generate"],
                    help="profiling runs (tools/prof_round.sh): only the edit timed region / only one attention leg, few launches")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # dry run of the N > 1 line on a box with fewer GPUs than ranks (tools / tests only): UCE_BENCH_SAME_DEVICE=1 puts every rank
    # on cuda:0 and UCE_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("UCE_BENCH_SAME_DEVICE") == "1":
        local = 0
        # several processes on ONE device: the one-launch GroupNorm waits grid-wide inside a sample and needs its whole grid resident -
        # with another process's grids on the same CUs that is not guaranteed (the wait is bounded, but two such kernels can starve
        # each other until it expires); the dry run takes the two-kernel form
        os.environ.setdefault("UCE_GN_FUSED", "0")
    backend = os.environ.get("UCE_BENCH_BACKEND", "nccl")
    dev_index = local if local < torch.cuda.device_count() else 0     # a launcher that masks one GPU per rank shows it as cuda:0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pinned_cores = 0
    if world > 1:
        from uce_amd import generate as _gen
        pinned_cores = _gen.pin_rank_to_cores(local, world)           # each rank on its own block of cores (its GPU's NUMA node first)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # the line says n_gpus = WORLD_SIZE: refuse to print it unless the job really is `--gpus` ranks in ONE process group, one
    # distinct GPU each (a launcher started with another rank count, or a group that came up smaller, must not pass for an N-GPU run)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: refusing to report an N-GPU line")
    if world > 1:
        import torch.distributed as dist
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        if backend == "nccl" and os.environ.get("UCE_BENCH_SAME_DEVICE") != "1":
            # (host, device index) - not the device UUID: a runtime that reports one UUID for every GPU of a node must not
            # turn a correct 8-GPU launch into a refusal; RCCL itself rejects two ranks on one device
            ids = [None] * world
            dist.all_gather_object(ids, (socket.gethostname(), int(local)))
            if len(set(ids)) != world:
                raise SystemExit(f"{world} ranks share {len(set(ids))} GPUs: refusing to report an N-GPU line")

    from uce_amd import edit as E
    from uce_amd import cli
    H = E.UceHandle.get(device)
    algo = cli.ALGO_IDS[args.algo]
    # CFG batch of the attention legs: that of the generation leg, capped at 128 - the batch the PMC passes of profiles/ were
    # collected at (tools/prof_round.sh), so every shape keeps its traffic entry
    gb = min(128, 2 * max(1, min(args.gen_batch, max(args.gen_images, 1))))
    if args.only == "xattn":            # fixed launch counts for the PMC passes (tools/pmc_fold.py splits by order)
        print(json.dumps(xattn_leg(device, (2, gb), iters=4)), flush=True)
        return
    if args.only == "sattn":
        print(json.dumps(sattn_leg(device, gb, iters=4, with_torch=False)), flush=True)
        return
    if args.only == "generate":         # the images/s leg alone (unedited synthetic weights), for A/B runs
        g = generation_leg(device, world, args.gen_images, args.gen_steps, None, args.gen_batch,
                           rowwise_images=args.gen_rowwise if world == 1 else min(args.gen_rowwise, 2))
        if rank == 0:
            print(json.dumps(g), flush=True)
        return

    r = run_edit(H, args.workload, device, args.steps, args.warmup, algo, world, breakdown=(rank == 0))
    inp, out, N = r["inp"], r["out"], r["N"]
    n_e, n_p, d, _, cfg_idx = WORKLOADS[args.workload]
    result = {
        "metric": "concepts/sec closed-form edit (SD-1.4, 768-d)",
        "value": round(world * N * args.steps / r["elapsed"], 1),
        "unit": "concepts/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(r["ms_per_step"], 5),
        "ms_per_step_events": round(r["ms_per_step_events"], 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (f64 Gram/solve)",
        "data": "synthetic",
        "host_cores_per_rank": pinned_cores if world > 1 else len(os.sched_getaffinity(0)),
        "config": {"workload": f"{args.workload}: {n_e} erase + {n_p} preserve concepts, d={d}, "
                               f"{len(inp['mods'])} attn2 to_k/to_v modules = one {inp['rows']}x{d} fp32 slab, "
                               f"lambda 0.5, algo {args.algo}",
                   "baseline_config": cfg_idx,
                   "parallelism": "replicas only" if world > 1 else "single GPU"},
    }
    if rank == 0:
        result["roofline"] = roofline_block(r)
    if args.only == "edit":
        if rank == 0:
            print(json.dumps(result), flush=True)
        return
    if rank == 0 and world == 1 and not args.no_configs:
        del r
        result["configs"] = []
        for name in CONFIG_LEGS:
            if name == args.workload:
                continue
            try:
                result["configs"].append(config_leg(H, name, device, algo))
            except Exception as err:  # noqa: BLE001
                result["configs"].append({"workload": name, "error": repr(err)})
    if args.gen_images > 0:
        kept = []
        result["generate"] = generation_leg(device, world, args.gen_images, args.gen_steps,
                                            out if out.shape[1] == 768 else None, args.gen_batch,
                                            rowwise_images=args.gen_rowwise if world == 1 else min(args.gen_rowwise, 2), keep_pipe=kept)
        if rank == 0 and world == 1 and kept and not args.no_configs:
            import tempfile
            try:
                with tempfile.TemporaryDirectory() as tmp:
                    result["uce_wall_s"] = uce_wall_leg(kept.pop(), device, tmp)
            except Exception as err:  # noqa: BLE001
                result["uce_wall_s"] = {"error": repr(err)}
        del kept
    if rank == 0 and args.gen_images > 0:
        result["xattn"] = xattn_leg(device, (2, gb))
        result["sattn"] = sattn_leg(device, gb)
    if world > 1:
        torch.distributed.barrier()                 # every timed region of every rank is over before rank 0 loads the host cores
    if rank == 0:
        _log("gpu part: " + json.dumps(result))
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cached_cpu_baseline(inp, world)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


CPU_BASELINE_CACHE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "uce_bench_cpu_baseline.json")


def _bench_sha() -> str:
    import hashlib
    return hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16]


CPU_BASELINE_MAX_AGE_S = 6 * 3600.0


def cached_cpu_baseline(inp, world: int):
    """N = 1: measure (and leave the figure in a scratch file).  N > 1: the driver runs N = 1, 2, 4, 8 back to back on one
    node - reuse the N = 1 measurement of this box when it is there (same host, same workload, same bench.py and kernel sources,
    younger than six hours), else take a shorter bounded sample now; either way the line says where and when the number was
    measured."""
    key = f"{inp['name']}|{cpu_model()}|{_bench_sha()}|{SOURCE_HASH}"
    if world > 1:
        try:
            c = json.load(open(CPU_BASELINE_CACHE))
            age = time.time() - float(c.get("measured_unix", 0))
            if c.get("key") == key and 0 <= age <= CPU_BASELINE_MAX_AGE_S:
                c["baseline"]["measured_at_n_gpus"] = 1
                c["baseline"]["cache_age_s"] = round(age, 1)
                return c["baseline"]
        except Exception:  # noqa: BLE001
            pass
    b = cpu_baseline(inp, repeats=5 if world == 1 else 2)
    b["measured_at_n_gpus"] = world
    b["cache_age_s"] = 0.0
    if world == 1:
        try:
            b["configs"] = cpu_baseline_configs("cpu", b["cores"])
        except Exception as err:  # noqa: BLE001
            b["configs"] = [{"error": repr(err)}]
        try:
            json.dump({"key": key, "measured_unix": time.time(), "baseline": b}, open(CPU_BASELINE_CACHE, "w"))
        except Exception:  # noqa: BLE001
            pass
    return b


if __name__ == "__main__":
    main()
