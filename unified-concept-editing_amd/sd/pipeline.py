"""Minimal Stable-Diffusion inference pipeline with the surface the reference scripts use:
    pipe.unet, pipe.tokenizer, pipe.encode_prompt(...), pipe.to(...), pipe(prompt, ...).images
(trainscripts/uce_sd_erase.py:29-39,197-200; evalscripts/generate-images-sd.py:13-19,37-42).

Weights: a local diffusers-format directory (`--model_dir`) when one exists, otherwise
seeded-random weights of the named architecture (`--synthetic_model`): the machines this runs on
have no network, no checkpoints and no tokenizer vocabulary, so throughput is measured on
synthetic weights and says so.  Numerical parity of full images with a real diffusers run is
UNPINNED; the pieces that are pinned are the closed-form edit (golden fixtures from
the reference itself) and the cross-attention kernel (torch SDPA goldens).
"""
from __future__ import annotations

import math
import os
import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .scheduler import EulerDiscreteScheduler, PNDMScheduler

# MIOpen times every applicable solver the first time it meets a convolution; its reference
# ("naive", f64-accumulating) solver is one of them and costs ~18 s of start-up per process at
# batch 2 (48 ms average over 384 calls in profiles/r01), far more at larger batches.  It never wins.
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
from . import unet as _unet
from .unet import UNet2DConditionModel, UNetConfig, conv2d, group_norm_act, linear, upsample2x_conv

MAX_LEN = 77
BOS, EOS = 49406, 49407


# ------------------------------------------------------------------------------------ tokenizer

class SyntheticTokenizer:
    """Stand-in for CLIPTokenizer when no vocabulary is on disk: one id per whitespace word
    (crc32 hash), BOS/EOS framing, EOS padding, truncation at 77 - the same call signature and the
    same attention_mask semantics the reference relies on (uce_sd_erase.py:34-39)."""
    model_max_length = MAX_LEN

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        L = max_length or MAX_LEN
        ids = np.full((len(texts), L), EOS, dtype=np.int64)         # (numpy rows, one tensor at the end: a torch.tensor + two indexed
        mask = np.zeros((len(texts), L), dtype=np.int64)            # assignments per string made 1 500 strings 68 ms of a UCE() call)
        crc = zlib.crc32
        for i, t in enumerate(texts):
            toks = [crc(w.encode("utf-8")) % 49000 + 256 for w in t.lower().split()][: L - 2]
            n = len(toks) + 2
            ids[i, 0] = BOS
            ids[i, 1:n - 1] = toks
            mask[i, :n] = 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}


def load_tokenizer(model_dir: Optional[str]):
    if model_dir and os.path.isdir(os.path.join(model_dir, "tokenizer")):
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer"))
    return SyntheticTokenizer()


# ------------------------------------------------------------------------------------ text encoder

@dataclass
class TextConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12

    @classmethod
    def tiny(cls, hidden: int = 64):
        return cls(hidden, hidden * 2, 2, 2)


def build_text_encoder(cfg: TextConfig, model_dir: Optional[str] = None, sub: str = "text_encoder",
                       projection_dim: Optional[int] = None, hidden_act: str = "quick_gelu"):
    """CLIPTextModel (SD-1.x, SDXL's first encoder) or, with `projection_dim`, CLIPTextModelWithProjection (SDXL's
    second, OpenCLIP-bigG: its pooled `text_embeds` feed the U-Net's micro-conditioning)."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    cls = CLIPTextModel if projection_dim is None else CLIPTextModelWithProjection
    if model_dir and os.path.isdir(os.path.join(model_dir, sub)):
        return cls.from_pretrained(os.path.join(model_dir, sub))
    kw = {} if projection_dim is None else {"projection_dim": projection_dim}
    c = CLIPTextConfig(vocab_size=49408, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                       num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                       max_position_embeddings=MAX_LEN, hidden_act=hidden_act, **kw)
    return cls(c)


# ------------------------------------------------------------------------------------ VAE decoder

class _VaeResnet(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        xs = x if self.conv_shortcut is None else conv2d(self.conv_shortcut, x)
        # the residual join rides in conv2's epilogue
        return conv2d(self.conv2, group_norm_act(self.norm2, conv2d(self.conv1, group_norm_act(self.norm1, x, True)), True),
                      residual=xs)


class _VaeAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = group_norm_act(self.group_norm, x, False).permute(0, 2, 3, 1).reshape(B, H * W, C)
        xr = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
        L = H * W
        if _unet.hip16(h) and C % 32 == 0 and L % 32 == 0 and L <= 16384 and self.to_v.weight.dtype == h.dtype:
            # ONE head of C = 512 dims: too wide for the register-resident attention kernels, so per image the two products
            # run on the linear kernel around a row softmax (scores in f32):  S = q k^T,  P = softmax(S / sqrt(C)),
            # O = P v + b_v (rows of P sum to one).  v^T comes straight out of its projection with the operands swapped
            # (v^T = W_v h^T), so nothing is transposed.
            from .. import edit as _edit
            hd = _edit.UceHandle.get(h.device)
            q, k = linear(self.to_q, h), linear(self.to_k, h)
            o = torch.empty_like(q)
            wv, bv = self.to_v.weight, self.to_v.bias
            for i in range(B):
                vt = hd.linear(wv, h[i])                                   # [C, L]
                p = hd.softmax_rows(hd.linear_f32(q[i], k[i]), C ** -0.5, h.dtype)
                hd.linear(p, vt, bv, out=o[i])
            return linear(self.to_out[0], o, residual=xr).reshape(B, H, W, C).permute(0, 3, 1, 2)
        if _unet.hip16(h):
            raise RuntimeError(f"VAE attention over {L} tokens of {C} channels ({h.dtype}) has no HIP path (C % 32 == 0, L % 32 == 0, "
                               "L <= 16384): there is no library attention behind the product path")
        o = F.scaled_dot_product_attention(linear(self.to_q, h)[:, None], linear(self.to_k, h)[:, None],
                                           linear(self.to_v, h)[:, None])[:, 0]
        return linear(self.to_out[0], o, residual=xr).reshape(B, H, W, C).permute(0, 3, 1, 2)


class _VaeUp(nn.Module):
    def __init__(self, cin, cout, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeResnet(cin if i == 0 else cout, cout) for i in range(3)])
        self.upsamplers = nn.ModuleList([_UpConv(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _UpConv(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return upsample2x_conv(self.conv, x)


class _VaeMid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([_VaeAttention(c)])
        self.resnets = nn.ModuleList([_VaeResnet(c, c), _VaeResnet(c, c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Decoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=4):
        super().__init__()
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.up_blocks = nn.ModuleList([])
        self.mid_block = _VaeMid(rev[0])
        prev = rev[0]
        for i, c in enumerate(rev):
            self.up_blocks.append(_VaeUp(prev, c, add_up=i < len(rev) - 1))
            prev = c
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(conv2d(self.conv_in, z))
        for u in self.up_blocks:
            x = u(x)
        return conv2d(self.conv_out, group_norm_act(self.conv_norm_out, x, True))


class VaeDecoder(nn.Module):
    """AutoencoderKL's decode path (post_quant_conv + decoder), scaling factor 0.18215."""
    scaling_factor = 0.18215

    def __init__(self, ch=(128, 256, 512, 512), scaling_factor: float = 0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor                 # SD-1.x 0.18215, SDXL 0.13025
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.decoder = _Decoder(ch)

    def decode(self, latents):
        return self.decoder(conv2d(self.post_quant_conv, latents / self.scaling_factor))


# ------------------------------------------------------------------------------------ pipeline

def _check_latent_size(pipe, hh: int, ww: int) -> None:
    """On a GPU in a 16-bit dtype every layer runs on the library's own kernels and a layer they cannot take raises - up front
    and by name here, not in the middle of a denoising loop: the U-Net halves the map three times (stride-2 convolutions, 2x
    upsamples back), so the latent height / width must be multiples of 8 (pixel sizes multiples of 64)."""
    if torch.device(pipe.device).type == "cuda" and getattr(pipe, "dtype", torch.float32) in (torch.bfloat16, torch.float16):
        if hh <= 0 or ww <= 0 or hh % 8 or ww % 8:
            raise ValueError(f"height x width = {hh * 8} x {ww * 8}: the 16-bit GPU path needs multiples of 64 pixels "
                             "(the U-Net halves the latent map three times); use a multiple of 64 or run the pipeline in fp32")


def _raise_on_expired_wait(device) -> None:
    """A kernel of the step that waits on other workgroups of its own launch (the one-launch GroupNorm) reports an expired wait
    through the handle's status word instead of hanging; its output is NaN then.  One 4-byte read per pipe() call - the call
    synchronises anyway to hand images / latents to the host - turns that into an exception instead of black PNGs."""
    if torch.device(device).type != "cuda":
        return
    from ..edit import UceHandle
    UceHandle.get(device).status()


def images_from_decoded(decoded: torch.Tensor, output_type: str) -> list:
    """diffusers' post-processing of the VAE output (`(image / 2 + 0.5).clamp(0, 1)`, NHWC float32; `numpy_to_pil`:
    `(images * 255).round().astype("uint8")`) - the arithmetic of the "pil" branch runs where the tensor is: the same float32
    multiply, round-half-to-even and conversion give the same bytes on the device, a quarter of the bytes cross PCIe, and
    the host does not spend a millisecond per image between two U-Net batches while the GPU idles."""
    img = (decoded.float() / 2 + 0.5).clamp(0, 1)
    if output_type == "pil":
        from PIL import Image
        u8 = (img * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()
        return [Image.fromarray(im) for im in u8]
    return list(img.permute(0, 2, 3, 1).cpu().numpy())


@dataclass
class PipeOutput:
    images: list
    latents: torch.Tensor


class _GraphedStep:
    """One denoising evaluation - cat(latents) -> U-Net -> classifier-free-guidance combine - captured
    once in a hipGraph and replayed per step.  At batch 2 the eager U-Net is launch-bound (~1 900
    launches of a few microseconds each per call); a replay issues them back to back.  Inputs live in
    fixed buffers: the latents, the timestep (device tensor), the text context and the hoisted K/V of
    every attn2 (rewritten per prompt by `set_context`).  Parameters are captured by address, so an
    in-place weight patch (`patch_unet`) is picked up without re-capturing."""

    def __init__(self, pipe, n: int, hh: int, ww: int, cfg: bool, guidance_scale: float, ctx: torch.Tensor,
                 raw: bool = False):
        from .unet import Attention, linear
        self._linear = linear
        unet, dev = pipe.unet, pipe.device
        self.unet = unet
        self.raw = raw
        self.lat = torch.zeros((n, unet.cfg.in_channels, hh, ww), device=dev, dtype=pipe.dtype)
        self.t = torch.zeros((1,), device=dev, dtype=torch.long)
        self.ctx = ctx.clone()
        self.cross = [m for m in unet.modules() if isinstance(m, Attention) and m.is_cross]
        # (the hoisted K / V through the SAME entry point as the eager path's cache_context - uce_linear_fwd, no library GEMM)
        self.kv = [(linear(m.to_k, self.ctx), linear(m.to_v, self.ctx)) for m in self.cross]
        self._bind()

        def body():
            x = torch.cat([self.lat] * 2) if cfg else self.lat
            eps = unet(x, self.t, self.ctx)
            if cfg and not raw:                            # raw: the guidance combine rides in the fused scheduler step
                eu, ec = eps.chunk(2)
                eps = eu + guidance_scale * (ec - eu)
            return eps

        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                      # warm-up off the capture: kernel selection, workspaces
            for _ in range(2):
                body()
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: other threads of the process (the RCCL watchdog of a multi-GPU run
        # polls events) must not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.eps = body()

    def _bind(self):
        for m, kv in zip(self.cross, self.kv):
            m.kv_cache = kv

    def set_context(self, ctx: torch.Tensor) -> None:
        self.ctx.copy_(ctx)
        for m, (k, v) in zip(self.cross, self.kv):
            k.copy_(self._linear(m.to_k, self.ctx))        # the same call as UNet.cache_context: graphed and eager runs give the same bits
            v.copy_(self._linear(m.to_v, self.ctx))
        self._bind()

    def __call__(self, latents: torch.Tensor, t: int) -> torch.Tensor:
        self.lat.copy_(latents)
        self.t.fill_(int(t))
        self.graph.replay()
        return self.eps if self.raw else self.eps.clone()  # PNDM keeps a history of past outputs (raw: consumed at once)


class StableDiffusionPipeline:
    def __init__(self, unet: UNet2DConditionModel, text_encoder, tokenizer, vae: Optional[VaeDecoder],
                 scheduler: Optional[PNDMScheduler] = None):
        self.unet, self.text_encoder, self.tokenizer, self.vae = unet, text_encoder, tokenizer, vae
        self.scheduler = scheduler or PNDMScheduler()
        self.device = torch.device("cpu")
        self.dtype = torch.float32
        self.hoist_context = True        # K/V of the (step-invariant) text context computed once per prompt
        self.use_graph = True            # on a GPU: replay the denoising evaluation from a hipGraph
        self.fused_step = True           # on a GPU in bf16/f16: guidance combine + PLMS step as one HIP launch
        self.channels_last = True        # on a GPU: NHWC weights/activations for the convolutions
        self._graphs: Dict[tuple, _GraphedStep] = {}

    # -- same call the reference makes: DiffusionPipeline.from_pretrained(...).to(device)
    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        self._graphs.clear()             # parameter storage may move: captured addresses are stale
        for m in (self.unet, self.text_encoder, self.vae):
            if m is not None:
                m.to(device=device, dtype=dtype)
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        if self.device.type == "cuda" and self.channels_last:
            # MIOpen's bf16 implicit-GEMM kernels are NHWC: NCHW activations cost two transposes per conv
            for m in (self.unet, self.vae):
                if m is not None:
                    m.to(memory_format=torch.channels_last)
        return self

    def set_progress_bar_config(self, **kw):
        pass

    @torch.no_grad()
    def encode_prompt(self, prompt, device=None, num_images_per_prompt: int = 1,
                      do_classifier_free_guidance: bool = False, negative_prompt=None, **kw):
        device = torch.device(device) if device is not None else self.device
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)

        def enc(texts):
            tok = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt")
            out = self.text_encoder(input_ids=tok["input_ids"].to(device))[0]
            return out.to(self.dtype).repeat_interleave(num_images_per_prompt, dim=0)

        pe = enc(prompts)
        ne = None
        if do_classifier_free_guidance:
            neg = [""] * len(prompts) if negative_prompt is None else (
                [negative_prompt] * len(prompts) if isinstance(negative_prompt, str) else list(negative_prompt))
            ne = enc(neg)
        return pe, ne

    def encode_prompt_prefix(self, prompts, device, n_pos: int, input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Hidden states of the FIRST `n_pos` token positions only -> [B, n_pos, d] (pipeline dtype); `input_ids` = what
        `self.tokenizer` returned for `prompts` when the caller has tokenised them already.  CLIP's text encoder is
        causal: position p attends to positions <= p, every other layer is row-wise, so the states of a prefix are those of the
        full 77-position forward (up to the summation order of a GEMM of another height) - and the closed-form edit reads ONE
        position per string, `attention_mask.sum() - 2` (uce_sd_erase.py:25-42), which for a concept name sits at position 2-8:
        edit.last_token_embeddings runs the encoder on the positions up to the batch's largest index instead of all 77."""
        device = torch.device(device) if device is not None else self.device
        if input_ids is None:
            input_ids = self.tokenizer(list(prompts), padding="max_length", max_length=self.tokenizer.model_max_length,
                                       truncation=True, return_tensors="pt")["input_ids"]
        return self.text_encoder(input_ids=input_ids[:, :n_pos].to(device))[0].to(self.dtype)

    def _draw_latents(self, n_prompts: int, n: int, hh: int, ww: int, generator) -> torch.Tensor:
        """diffusers' randn_tensor: a CPU generator draws on the CPU in the target dtype, then moves.
        One generator -> one draw of the whole batch (what generate-images-sd.py:37-42 gets for its single
        prompt).  A list with one generator per PROMPT -> each draws its prompt's [n, C, h, w] block, so a
        batch of prompts reproduces exactly the latents the reference draws prompt by prompt; a list with one
        generator per IMAGE is diffusers' own list form ([1, C, h, w] each)."""
        C = self.unet.cfg.in_channels

        def draw(g, k):
            gdev = g.device if g is not None else self.device
            return torch.randn((k, C, hh, ww), generator=g, device=gdev, dtype=self.dtype).to(self.device)

        if isinstance(generator, (list, tuple)):
            if len(generator) == n_prompts:
                return torch.cat([draw(g, n) for g in generator])
            if len(generator) == n_prompts * n:
                return torch.cat([draw(g, 1) for g in generator])
            raise ValueError(f"got {len(generator)} generators for {n_prompts} prompts x {n} images")
        return draw(generator, n_prompts * n)

    @torch.no_grad()
    def __call__(self, prompt, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 num_images_per_prompt: int = 1, generator=None,
                 output_type: str = "pil", height: int = None, width: int = None, callback=None,
                 latents: Optional[torch.Tensor] = None, **kw) -> PipeOutput:
        """`prompt` may be a list: the prompts are denoised as ONE batch (images are independent units, so
        this is the same result per image as calling prompt by prompt, at a multiple of the throughput)."""
        n = num_images_per_prompt
        cfg = guidance_scale > 1.0
        n_prompts = 1 if isinstance(prompt, str) else len(prompt)
        pe, ne = self.encode_prompt(prompt, self.device, n, cfg)
        ctx = torch.cat([ne, pe]) if cfg else pe
        s = self.unet.cfg.sample_size
        hh, ww = (height // 8 if height else s), (width // 8 if width else s)
        _check_latent_size(self, hh, ww)
        if latents is None:
            latents = self._draw_latents(n_prompts, n, hh, ww, generator)
        else:                                              # diffusers' `latents=`: pre-drawn initial noise
            latents = latents.to(device=self.device, dtype=self.dtype)
        n = n_prompts * n                                  # batch of the denoising loop from here on
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, device="cpu")
        latents = latents * sch.init_noise_sigma
        # guidance combine + scheduler step in one launch (uce_cfg_pndm_step) where the tensors allow it
        fused = (self.fused_step and callback is None and self.device.type == "cuda" and isinstance(sch, PNDMScheduler)
                 and self.dtype in (torch.bfloat16, torch.float16) and latents.numel() % 8 == 0)
        handle = None
        if fused:
            from .. import edit as _edit
            handle = _edit.UceHandle.get(self.device)
        graphed = None
        if self.use_graph and self.hoist_context and self.device.type == "cuda":
            key = (n, hh, ww, cfg, float(guidance_scale), tuple(ctx.shape), self.dtype, fused)
            graphed = self._graphs.get(key)
            if graphed is None:
                try:
                    graphed = self._graphs[key] = _GraphedStep(self, n, hh, ww, cfg, float(guidance_scale), ctx, raw=fused)
                except RuntimeError as err:               # capture refused: same kernels, eager launches
                    import warnings
                    warnings.warn(f"hipGraph capture of the denoising step failed ({err}); launching eagerly")
                    self.use_graph = False
                    torch.cuda.synchronize(self.device)
            if graphed is not None:
                graphed.set_context(ctx)
        if graphed is None and self.hoist_context:
            self.unet.cache_context(ctx)
        try:
            for step_index, t in enumerate(sch.timesteps.tolist()):
                if graphed is not None:
                    eps = graphed(latents, t)
                else:
                    x = torch.cat([latents] * 2) if cfg else latents
                    eps = self.unet(x, torch.tensor([t], device=self.device), ctx)
                    if cfg and not fused:
                        eu, ec = eps.chunk(2)
                        eps = eu + guidance_scale * (ec - eu)
                if fused:
                    latents = sch.step_fused(eps, cfg, float(guidance_scale), t, latents, handle)
                    continue
                if callback is not None:                   # (step index, timestep, latents fed to the U-Net, guided eps)
                    callback(step_index, t, latents, eps)
                latents = sch.step(eps, t, latents)
        finally:
            self.unet.cache_context(None)
        images: list = []
        if output_type != "latent" and self.vae is not None:
            images = images_from_decoded(self.vae.decode(latents), output_type)
        _raise_on_expired_wait(self.device)
        return PipeOutput(images=images, latents=latents)


class StableDiffusionXLPipeline(StableDiffusionPipeline):
    """The surface uce_sd_debias.py uses of diffusers' StableDiffusionXLPipeline (`--model_id
    stabilityai/stable-diffusion-xl-base-1.0`, uce_sd_debias.py:240-242): `.unet` (140 attn2 projections, 2048-d
    context), `.tokenizer` (CLIP-L's), `.encode_prompt(...)[0]` = the two encoders' penultimate hidden states
    concatenated ([B, 77, 768 + 1280]), `pipe(prompt, ...).images` with the Euler scheduler, the pooled-text +
    size/crop micro-conditioning and 1024x1024 default size.  Restated from the published diffusers==0.33.0
    pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py; numerics against diffusers are unpinned."""

    def __init__(self, unet, text_encoder, text_encoder_2, tokenizer, tokenizer_2, vae, scheduler=None):
        super().__init__(unet, text_encoder, tokenizer, vae, scheduler or EulerDiscreteScheduler())
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.use_graph = False             # the captured step of the base class is the SD-1.x call; SDXL launches eagerly
        self.force_zeros_for_empty_prompt = True

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        super().to(device, dtype)
        self.text_encoder_2.to(device=device, dtype=dtype)
        return self

    @torch.no_grad()
    def encode_prompt(self, prompt, device=None, num_images_per_prompt: int = 1,
                      do_classifier_free_guidance: bool = False, negative_prompt=None, **kw):
        """-> (prompt_embeds [B, 77, 2048], negative_prompt_embeds, pooled [B, 1280], negative_pooled)."""
        device = torch.device(device) if device is not None else self.device
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)

        def enc(texts):
            parts, pooled = [], None
            for tok, te in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)):
                ids = tok(texts, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="pt")["input_ids"].to(device)
                out = te(input_ids=ids, output_hidden_states=True)
                pooled = out[0]                                   # kept from the LAST encoder: its projected text_embeds
                parts.append(out.hidden_states[-2])
            pe = torch.cat(parts, dim=-1).to(self.dtype)
            return (pe.repeat_interleave(num_images_per_prompt, dim=0),
                    pooled.to(self.dtype).repeat_interleave(num_images_per_prompt, dim=0))

        pe, pp = enc(prompts)
        ne = npool = None
        if do_classifier_free_guidance:
            if negative_prompt is None and self.force_zeros_for_empty_prompt:
                ne, npool = torch.zeros_like(pe), torch.zeros_like(pp)
            else:
                neg = [""] * len(prompts) if negative_prompt is None else (
                    [negative_prompt] * len(prompts) if isinstance(negative_prompt, str) else list(negative_prompt))
                ne, npool = enc(neg)
        return pe, ne, pp, npool

    def encode_prompt_prefix(self, prompts, device, n_pos: int, input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The first `n_pos` positions of `prompt_embeds` (both encoders' penultimate states, causal: see StableDiffusionPipeline);
        `input_ids`: the first tokenizer's, when the caller has them."""
        device = torch.device(device) if device is not None else self.device
        parts = []
        for tok, te in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)):
            ids = input_ids if (input_ids is not None and tok is self.tokenizer) else tok(
                list(prompts), padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")["input_ids"]
            ids = ids[:, :n_pos].to(device)
            parts.append(te(input_ids=ids, output_hidden_states=True).hidden_states[-2])
        return torch.cat(parts, dim=-1).to(self.dtype)

    @torch.no_grad()
    def __call__(self, prompt, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 num_images_per_prompt: int = 1, generator=None, output_type: str = "pil", height: int = None,
                 width: int = None, callback=None, latents: Optional[torch.Tensor] = None, **kw) -> PipeOutput:
        n = num_images_per_prompt
        cfg = guidance_scale > 1.0
        n_prompts = 1 if isinstance(prompt, str) else len(prompt)
        pe, ne, pp, npool = self.encode_prompt(prompt, self.device, n, cfg)
        s = self.unet.cfg.sample_size
        height, width = height or s * 8, width or s * 8
        hh, ww = height // 8, width // 8
        _check_latent_size(self, hh, ww)
        if latents is None:
            latents = self._draw_latents(n_prompts, n, hh, ww, generator)
        else:
            latents = latents.to(device=self.device, dtype=self.dtype)
        n = n_prompts * n
        ids = torch.tensor([[height, width, 0, 0, height, width]], dtype=self.dtype, device=self.device).repeat(n, 1)
        ctx = torch.cat([ne, pe]) if cfg else pe
        added = {"text_embeds": torch.cat([npool, pp]) if cfg else pp, "time_ids": torch.cat([ids, ids]) if cfg else ids}
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, device="cpu")
        latents = latents * sch.init_noise_sigma
        if self.hoist_context:
            self.unet.cache_context(ctx)
        try:
            for step_index, t in enumerate(sch.timesteps.tolist()):
                x = torch.cat([latents] * 2) if cfg else latents
                x = sch.scale_model_input(x, t)
                eps = self.unet(x, torch.tensor([t], device=self.device), ctx, added_cond_kwargs=added)
                if cfg:
                    eu, ec = eps.chunk(2)
                    eps = eu + guidance_scale * (ec - eu)
                if callback is not None:
                    callback(step_index, t, latents, eps)
                latents = sch.step(eps, t, latents)
        finally:
            self.unet.cache_context(None)
        images: list = []
        if output_type != "latent" and self.vae is not None:
            images = images_from_decoded(self.vae.decode(latents), output_type)
        _raise_on_expired_wait(self.device)
        return PipeOutput(images=images, latents=latents)


# ------------------------------------------------------------------------------------ loading

ARCH = {
    "CompVis/stable-diffusion-v1-4": ("sd1", 768),
    "runwayml/stable-diffusion-v1-5": ("sd1", 768),
    "stabilityai/stable-diffusion-xl-base-1.0": ("sdxl", 2048),
    "tiny-sd-test": ("tiny", 64),
    "tiny-sdxl-test": ("tiny_xl", 64),
}


def _seeded_init(module: nn.Module, seed: int) -> None:
    g = torch.Generator().manual_seed(seed)
    for p in module.parameters():
        with torch.no_grad():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / math.sqrt(fan_in))
            # 1-D parameters (norm weights 1 / biases 0) keep their defaults


def load_pipeline(model_id: str, torch_dtype=torch.float32, device="cpu", model_dir: Optional[str] = None,
                  synthetic: bool = False, vae: bool = True, seed: int = 0) -> StableDiffusionPipeline:
    """`DiffusionPipeline.from_pretrained(model_id, ...)` replacement.  Order: real diffusers if
    importable and not synthetic -> local directory -> seeded-random weights (only on request)."""
    if not synthetic and model_dir is None:
        try:  # a machine that has diffusers + weights: use the real thing, the edit/generate code is agnostic
            from diffusers import DiffusionPipeline  # type: ignore
            kw = dict(torch_dtype=torch_dtype, safety_checker=None)
            if not vae:
                kw["vae"] = None
            return DiffusionPipeline.from_pretrained(model_id, **kw).to(device)
        except ImportError:
            raise RuntimeError(
                f"cannot load '{model_id}': diffusers is not installed and no --model_dir was given. "
                "Pass --model_dir <diffusers-format directory> or --synthetic_model (random weights).")
    if model_id not in ARCH:
        raise ValueError(f"unknown architecture '{model_id}': this runtime builds {sorted(ARCH)} "
                         "(install diffusers to load any other hub id)")
    kind, _ = ARCH[model_id]
    if kind in ("sdxl", "tiny_xl"):
        return _load_sdxl(kind, torch_dtype, device, model_dir, vae, seed)
    ucfg = UNetConfig.tiny() if kind == "tiny" else UNetConfig.sd14()
    tcfg = TextConfig.tiny(ucfg.cross_attention_dim) if kind == "tiny" else TextConfig()
    torch.manual_seed(seed)
    unet = UNet2DConditionModel(ucfg)
    text = build_text_encoder(tcfg, model_dir)
    vae_m = VaeDecoder((32, 32, 64, 64) if kind == "tiny" else (128, 256, 512, 512)) if vae else None
    if model_dir:
        from safetensors.torch import load_file
        up = os.path.join(model_dir, "unet", "diffusion_pytorch_model.safetensors")
        unet.load_state_dict(load_file(up), strict=True)
        vp = os.path.join(model_dir, "vae", "diffusion_pytorch_model.safetensors")
        if vae_m is not None and os.path.exists(vp):
            sd = {convert_deprecated_vae_key(k): v for k, v in load_file(vp).items()
                  if k.startswith(("decoder.", "post_quant_conv."))}
            for k, v in sd.items():       # SD-1.x checkpoints store the attention projections as 1x1 convolutions
                if ".attentions." in k and k.endswith(".weight") and v.dim() == 4:
                    sd[k] = v[:, :, 0, 0]
            vae_m.load_state_dict(sd, strict=True)
    for m in (unet, text, vae_m):
        if m is not None:
            m.eval().requires_grad_(False)
    pipe = StableDiffusionPipeline(unet, text, load_tokenizer(model_dir), vae_m)
    return pipe.to(device, torch_dtype)


def _load_sdxl(kind: str, torch_dtype, device, model_dir: Optional[str], vae: bool, seed: int) -> StableDiffusionXLPipeline:
    """SDXL-base (or its small-width test twin): U-Net with 140 attn2 projections, CLIP-L + OpenCLIP-bigG text encoders,
    the SDXL VAE scaling.  Random weights are drawn directly on the target device (2.6 G parameters)."""
    tiny = kind == "tiny_xl"
    ucfg = UNetConfig.tiny_xl() if tiny else UNetConfig.sdxl()
    t1 = TextConfig.tiny(32) if tiny else TextConfig()
    t2 = TextConfig.tiny(32) if tiny else TextConfig(1280, 5120, 32, 20)
    torch.manual_seed(seed)
    dev = torch.device(device)
    with torch.device(dev if (dev.type == "cuda" and not model_dir) else "cpu"):
        unet = UNet2DConditionModel(ucfg)
        text1 = build_text_encoder(t1, model_dir)
        text2 = build_text_encoder(t2, model_dir, "text_encoder_2", projection_dim=32 if tiny else 1280, hidden_act="gelu")
        vae_m = VaeDecoder((32, 32, 64, 64) if tiny else (128, 256, 512, 512), scaling_factor=0.13025) if vae else None
    if model_dir:
        from safetensors.torch import load_file
        unet.load_state_dict(load_file(os.path.join(model_dir, "unet", "diffusion_pytorch_model.safetensors")), strict=True)
        vp = os.path.join(model_dir, "vae", "diffusion_pytorch_model.safetensors")
        if vae_m is not None and os.path.exists(vp):
            sd = {convert_deprecated_vae_key(k): v for k, v in load_file(vp).items()
                  if k.startswith(("decoder.", "post_quant_conv."))}
            vae_m.load_state_dict(sd, strict=True)
    for m in (unet, text1, text2, vae_m):
        if m is not None:
            m.eval().requires_grad_(False)
    tok2 = SyntheticTokenizer()
    if model_dir and os.path.isdir(os.path.join(model_dir, "tokenizer_2")):
        from transformers import CLIPTokenizer
        tok2 = CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer_2"))
    pipe = StableDiffusionXLPipeline(unet, text1, text2, load_tokenizer(model_dir), tok2, vae_m)
    return pipe.to(device, torch_dtype)


_DEPRECATED_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def convert_deprecated_vae_key(key: str) -> str:
    """SD-1.x VAE checkpoints name the mid-block attention `query/key/value/proj_attn`; diffusers renames them to
    `to_q/to_k/to_v/to_out.0` at load time (`_convert_deprecated_attention_blocks`).  Same mapping here, so that a
    real checkpoint never leaves the attention randomly initialised."""
    if ".attentions." not in key:
        return key
    head, _, leaf = key.rpartition(".")
    stem, _, name = head.rpartition(".")
    return f"{stem}.{_DEPRECATED_VAE_ATTN[name]}.{leaf}" if name in _DEPRECATED_VAE_ATTN else key


def patch_unet(pipe, state: Dict[str, torch.Tensor]) -> List[str]:
    """`pipe.unet.load_state_dict(uce_weights, strict=False)` (generate-images-sd.py:17-19), with the
    fp32 -> bf16 cast done by the HIP kernel when the parameters live on a GPU in bf16."""
    params = dict(pipe.unet.named_parameters())
    handle = None
    loaded = []
    stale = False
    for k, v in state.items():
        p = params.get(k)
        if p is None:                     # strict=False: keys the U-Net does not have are ignored
            continue
        loaded.append(k)
        if tuple(p.shape) != tuple(v.shape):
            raise ValueError(f"{k}: shape {tuple(v.shape)} vs parameter {tuple(p.shape)}")
        if p.is_cuda and p.dtype == torch.bfloat16 and v.dtype == torch.float32:
            from .. import edit as _edit
            handle = handle or _edit.UceHandle.get(p.device)
            handle.cast_bf16(v.to(p.device).contiguous(), p.data)
        else:
            p.data.copy_(v.to(device=p.device, dtype=p.dtype))
        # both writes above go around the tensor's version counter: tensors derived from the old values (packed q|k|v rows,
        # interleaved GEGLU rows, the stacked time projections, zero-padded narrow weights - sd/unet.py `derived`) are stale,
        # and so is a captured step that holds their addresses.  The edited projections themselves (attn2.to_k / to_v) have
        # nothing derived from them: the usual patch keeps the captured graphs.
        if not (".attn2.to_k." in "." + k or ".attn2.to_v." in "." + k):
            stale = True
    if stale:
        from .unet import clear_derived
        if clear_derived(pipe.unet) and hasattr(pipe, "_graphs"):
            pipe._graphs.clear()
    return loaded
