"""Image-generation eval path: the drop-in for `generate_images` of the reference's
evalscripts/generate-images-sd.py:10-46, prompt-sharded over the GPUs of one node.

Reference behaviour kept: bf16 pipeline, optional patch of the U-Net with the edited attn2 weights
(`load_state_dict(strict=False)`, :17-19), one `pipe(...)` call per CSV row with
`generator=torch.Generator().manual_seed(evaluation_seed)` (a CPU generator, :41), rows filtered by
`from_case <= case_number <= till_case` (:33), PNGs named `{save_path}/{exp_name}/{case}_{i}.png`.

Added: when launched with WORLD_SIZE > 1 (one process per GPU, torch.distributed; backend nccl =
RCCL over xGMI on ROCm, gloo on CPU) rank 0 reads the edited-weight file and BROADCASTS the blob
to the other ranks (one collective, ~77 MB fp32 for SD-1.4), then rank r generates the selected
rows with `index % world == r`: no data-path collective after that.  `--batch_prompts B` denoises B
rows per U-Net call (at batch 2 = one prompt's CFG pair the U-Net is launch-bound on an MI355X; rows are
independent, so batching is free throughput) and PNG encoding overlaps the next batch on worker threads.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import torch

from .sd import pipeline as sdp


def dist_env() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(device: torch.device) -> Tuple[int, int]:
    rank, world, _ = dist_env()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = "nccl" if device.type == "cuda" else "gloo"
            kw = {"device_id": device} if device.type == "cuda" else {}
            dist.init_process_group(backend, **kw)
    return rank, world


def gpu_numa_node(local: int) -> int:
    """NUMA node of GPU `local` from sysfs (/sys/bus/pci/devices/<domain:bus:device.0>/numa_node), -1 when the platform does not say."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{int(getattr(pr, 'pci_domain_id', 0)):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
        return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except Exception:  # noqa: BLE001
        return -1


def rank_core_block(local: int, world: int, allowed=None, node: int = None):
    """The cores rank `local` of `world` (one process per GPU, one node) keeps to: a contiguous slice of the cores of its GPU's NUMA
    node - the node's cores divided among the ranks whose GPUs hang off it, assuming the GPUs are spread evenly over the nodes -
    or, when the node is unknown, an even split of every core the process may run on."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    node = gpu_numa_node(local) if node is None else node
    block = []
    if node >= 0:
        try:
            cpus = []
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
            ok = set(allowed)
            cpus = [c for c in cpus if c in ok]
            nodes = max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]))
            per_node = max(1, -(-world // nodes))                       # ranks sharing this node
            step = max(1, len(cpus) // per_node)
            k = local % per_node
            block = cpus[k * step:(k + 1) * step]
        except Exception:  # noqa: BLE001
            block = []
    if not block:
        step = max(1, len(allowed) // max(1, world))
        block = allowed[local * step:(local + 1) * step] or allowed
    return block


def pin_rank_to_cores(local: int, world: int) -> int:
    """One process per GPU on one node: the rank keeps to its own block of cores (`rank_core_block`: its GPU's NUMA node first), so
    that eight ranks' host work (CPU-seeded latent draws, tokenizer, graph replays, PNG encoding on worker threads) never migrates
    across sockets or piles onto the same cores.  Call it before any worker thread exists (they inherit the mask).  Returns the
    number of cores of the block (0: nothing changed - one rank, or a platform without sched_setaffinity); torch's intra-op pool
    is sized to the block."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return 0
    try:
        block = rank_core_block(local, world)
        os.sched_setaffinity(0, set(block))
        torch.set_num_threads(max(1, min(len(block), 8)))
        return len(block)
    except Exception:  # noqa: BLE001
        return 0


PNG_WORKERS_AUTO = -1


def png_worker_count(requested: int, world: int) -> int:
    """PNG-encoding threads of one rank.  `requested` < 0 (the default): a quarter of the cores the process may run on (after
    pin_rank_to_cores: the rank's own block), at most 32 - an image costs ~30 ms of one core, so eight threads keep up with the
    GPU many times over, but the files of the LAST batch are encoded with nothing left to hide behind: 128 images on 8 threads
    are 0.46 s, on 32 threads 0.12 s.  An explicit number is kept, capped so that the workers + the rank's own thread never ask
    for more threads than those cores; 0 = encode inline like the reference's loop."""
    try:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    except Exception:  # noqa: BLE001
        cores = os.cpu_count() or 8
    if int(requested) < 0:
        requested = max(1, min(32, cores // 4))
    return max(0, min(int(requested), max(1, cores - 1)))


def broadcast_uce_weights(path: Optional[str], keys_like, device: torch.device, rank: int, world: int,
                          force_collective: bool = False) -> Optional[Dict[str, torch.Tensor]]:
    """Rank 0 loads the safetensors artifact; everyone ends up with the same {name: fp32 tensor}.
    The tensors travel as ONE flat fp32 buffer (a single broadcast).  `force_collective` issues the broadcast even
    in a one-rank group (single-GPU smoke test of the RCCL path)."""
    if path is None:
        return None
    import torch.distributed as dist
    from safetensors.torch import load_file
    if world == 1 and not (force_collective and dist.is_initialized()):
        return load_file(path)
    meta: List = [None]
    state = None
    if rank == 0:
        state = load_file(path)
        meta = [[(k, list(v.shape)) for k, v in state.items()]]
    dist.broadcast_object_list(meta, src=0)
    layout = meta[0]
    total = sum(int(torch.Size(s).numel()) for _, s in layout)
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == 0:
        off = 0
        for k, shp in layout:
            n = state[k].numel()
            flat[off:off + n].copy_(state[k].reshape(-1))
            off += n
    dist.broadcast(flat, src=0)
    out, off = {}, 0
    for k, shp in layout:
        n = int(torch.Size(shp).numel())
        out[k] = flat[off:off + n].view(shp)
        off += n
    return out


def select_rows(df, from_case: int, till_case: int, rank: int, world: int):
    """(index, row) pairs this rank generates: the reference's case filter, then round-robin."""
    picked = [(i, r) for i, (_, r) in enumerate(df.iterrows())
              if (r.case_number >= from_case and r.case_number <= till_case)]
    return [(i, r) for j, (i, r) in enumerate(picked) if j % world == rank]


# activation + workspace footprint of one 512 x 512 image in flight through the build's own pipeline (CFG pair through the U-Net,
# VAE decode, PNG staging), rounded up from the bench's 128 prompts per call on a 288 GB MI355X; the automatic batch keeps half of
# the free HBM untouched
AUTO_BATCH_BYTES_PER_IMAGE = 3 << 29          # 1.5 GB
AUTO_BATCH_MAX_IMAGES = 128
# the automatic batch is rounded DOWN to this ladder: the tile form and the split of a layer's contraction depend on the rows of its
# GEMM, so the bf16 bits of an image depend on the batch it was denoised in - a batch that followed the free HBM byte for byte would
# make a fixed seed's file depend on the GPU's other tenants; on the ladder a given machine state maps to few, stable sizes
AUTO_BATCH_LADDER = (1, 2, 4, 8, 16, 32, 64, 128)


def auto_batch_prompts(pipe, dev: torch.device, num_images_per_prompt: int, n_rows: int) -> int:
    """Rows per `pipe()` call when the caller left `--batch_prompts` at its default (0): rows are independent (each keeps its own
    CPU-seeded latents, the draw of generate-images-sd.py:36,41), so the build's own pipeline on a GPU denoises as many per call as
    half of the free HBM holds, at most 128 images - at one row per call (the reference's loop) the U-Net's layers have too few
    output tiles to fill 256 CUs (2.2-3 images/s against 9.5 batched).  A foreign pipeline object (real diffusers, a test double)
    and CPU runs keep the reference's one row per call."""
    if dev.type != "cuda" or not isinstance(pipe, sdp.StableDiffusionPipeline) or n_rows <= 1:
        return 1
    free, _ = torch.cuda.mem_get_info(dev)
    side = getattr(getattr(pipe.unet, "cfg", None), "sample_size", 64) / 64.0          # (SDXL: 128 x 128 latents, 4x the pixels)
    images = max(1, min(AUTO_BATCH_MAX_IMAGES, int(free // 2 // int(AUTO_BATCH_BYTES_PER_IMAGE * side * side))))
    rows = max(1, images // max(1, int(num_images_per_prompt)))
    rows = max(b for b in AUTO_BATCH_LADDER if b <= rows)
    return max(1, min(n_rows, rows))


def generate_images(model_id, uce_model_path, prompts_path, save_path, exp_name="test", device="cuda:0",
                    torch_dtype=torch.bfloat16, guidance_scale=7.5, num_inference_steps=100,
                    num_images_per_prompt=10, from_case=0, till_case=1000000, model_dir=None, synthetic=False,
                    latents_only=False, skip_existing=False, pipe=None, batch_prompts: int = 0,
                    png_workers: int = PNG_WORKERS_AUTO) -> Dict[str, float]:
    """evalscripts/generate-images-sd.py:10-46.  `batch_prompts` CSV rows are denoised as one batch (each row
    still draws its latents from its own CPU generator seeded with `evaluation_seed`, exactly the draw the
    reference makes row by row); file names are those of the row-by-row loop, the contents of an image equal the row-alone result
    up to the bf16 summation order of the layers (tile forms depend on the batch: tests/test_generate_gpu.py states the distance).
    0 (the default) picks the batch from the free HBM on a fixed ladder (`auto_batch_prompts`) and halves it when a call runs out
    of memory; 1 is the reference's loop."""
    import pandas as pd
    rank, world, local = dist_env()
    dev = torch.device(device)
    if world > 1 and dev.type == "cuda":
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    rank, world = init_distributed(dev)
    if world > 1 and dev.type == "cuda":
        pin_rank_to_cores(local, world)                               # (before the worker threads exist: they inherit the mask)
    png_workers = png_worker_count(png_workers, world)

    # 1. the pipeline (every rank builds its own replica)
    if pipe is None:
        pipe = sdp.load_pipeline(model_id, torch_dtype=torch_dtype, device=dev, model_dir=model_dir,
                                 synthetic=synthetic, vae=not latents_only)
    # 2. edited weights: one broadcast from rank 0, then a by-name patch of the attn2 projections
    state = broadcast_uce_weights(uce_model_path, None, dev, rank, world)
    if state is not None:
        if hasattr(pipe.unet, "cfg"):
            sdp.patch_unet(pipe, state)
        else:  # a real diffusers pipeline
            pipe.unet.load_state_dict(state, strict=False)

    df = pd.read_csv(prompts_path)
    folder_path = f"{save_path}/{exp_name}"
    os.makedirs(folder_path, exist_ok=True)
    mine = select_rows(df, from_case, till_case, rank, world)

    t0 = time.perf_counter()
    n_img = 0
    todo = [row for _, row in mine
            if not (skip_existing and os.path.exists(f"{folder_path}/{row.case_number}_0.png"))]
    batch_prompts = batch_prompts_arg = int(batch_prompts)
    if batch_prompts <= 0:
        batch_prompts = auto_batch_prompts(pipe, dev, num_images_per_prompt, len(todo))
    # PNG encoding (host, ~30 ms per 512x512 image) runs on worker threads behind the next batch's denoising (eight by default, fewer
    # when the rank's block of cores is smaller: the tail after the LAST batch is batch x 30 ms / workers)
    writer = ThreadPoolExecutor(max_workers=png_workers) if (png_workers > 0 and not latents_only) else None
    pending = []
    auto_sized = batch_prompts_arg <= 0
    lo = 0
    while lo < len(todo):
        rows = todo[lo:lo + batch_prompts]
        prompts = [str(r.prompt) for r in rows]
        single = len(rows) == 1
        try:
            gens = [torch.Generator().manual_seed(int(r.evaluation_seed)) for r in rows]   # generate-images-sd.py:36
            out = pipe(prompt=prompts[0] if single else prompts, num_inference_steps=num_inference_steps,
                       guidance_scale=guidance_scale, num_images_per_prompt=num_images_per_prompt,
                       generator=gens[0] if single else gens,
                       **({"output_type": "latent"} if latents_only else {}))
        except torch.cuda.OutOfMemoryError:
            # the 1.5 GB per image of the automatic size is an estimate (another tenant, a foreign resolution): halve and retry
            # the same rows - their generators are re-seeded above, nothing of the failed call is kept
            if not auto_sized or batch_prompts <= 1:
                raise
            batch_prompts = max(1, batch_prompts // 2)
            torch.cuda.empty_cache()
            continue
        lo += len(rows)
        for i, r in enumerate(rows):
            sl = slice(i * num_images_per_prompt, (i + 1) * num_images_per_prompt)
            if latents_only:
                lat = out.latents if hasattr(out, "latents") else out.images
                torch.save(lat[sl].cpu(), f"{folder_path}/{r.case_number}.pt")
            else:
                for num, im in enumerate(out.images[sl]):
                    path = f"{folder_path}/{r.case_number}_{num}.png"
                    if writer is None:
                        im.save(path)
                    else:
                        pending.append(writer.submit(im.save, path))
        n_img += num_images_per_prompt * len(rows)
    for f in pending:
        f.result()                                            # surface encode / disk errors
    if writer is not None:
        writer.shutdown()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    stats = {"images": float(n_img), "seconds": elapsed, "rank": float(rank), "world": float(world),
             "batch_prompts": float(batch_prompts)}
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([float(n_img), elapsed], dtype=torch.float64, device=dev)
        tot = t.clone()
        dist.all_reduce(tot[:1], op=dist.ReduceOp.SUM)
        mx = t.clone()
        dist.all_reduce(mx[1:], op=dist.ReduceOp.MAX)
        stats["images_total"], stats["seconds_max"] = float(tot[0]), float(mx[1])
    return stats
