"""Host side of the closed-form UCE edit: the drop-in for `UCE()` of the reference's
trainscripts/uce_sd_erase.py:12-91 and trainscripts/uce_sd_debias.py:37-149.

What stays in Python (plumbing): module discovery by name, text-embedding extraction through
whatever pipeline object is handed in, packing every attn2.to_k/to_v weight into ONE [rows, d]
fp32 slab in HBM, the safetensors artifact.  What runs in hand-written HIP through the C ABI
(include/uce_hip.h): everything the reference does at :45-82 - Gram accumulation, the SPD
solve and the weight update - once for all modules instead of per module.

There is no CPU fallback: without libuce_hip.so / a GPU these functions raise.
"""
from __future__ import annotations

import ctypes
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib

torch.set_grad_enabled(False)


# --------------------------------------------------------------------------------------------
# handle + thin wrappers over the C ABI (device tensors in, device tensors out)
# --------------------------------------------------------------------------------------------

def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


from .sd.conv_dispatch import CONV_COLS_BYTES, conv_takes_igemm, even_chunk  # noqa: E402  (rule of the convolution wrappers below)


class UceHandle:
    """One per GPU.  Owns the library workspace (uce_create / uce_destroy)."""

    _cache: Dict[int, "UceHandle"] = {}

    def __init__(self, device: torch.device | str | int = "cuda:0"):
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError(f"uce_amd runs on an MI355X (device 'cuda:N' under ROCm), not on {self.device}")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the UCE edit has no CPU path")
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.index)
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self.lib.uce_create(ctypes.byref(h), self.index), "uce_create")
        self._h = h

    @classmethod
    def get(cls, device) -> "UceHandle":
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if idx not in cls._cache:
            cls._cache[idx] = cls(torch.device("cuda", idx))
        return cls._cache[idx]

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.uce_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- one wrapper per entry point -------------------------------------------------------
    def reserve(self, d_max: int, n_max: int) -> None:
        _lib.check(self.lib.uce_reserve(self._h, d_max, n_max), "uce_reserve")

    def gram(self, C: torch.Tensor, G: Optional[torch.Tensor], s: torch.Tensor, lamb: float):
        N, d = C.shape
        Ne = 0 if G is None else G.shape[0]
        A = torch.empty(d, d, dtype=torch.float64, device=self.device)
        Bt = torch.empty(d, d, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.uce_gram(self._h, _ptr(C), _ptr(G), _ptr(s), N, Ne, d, float(lamb), _ptr(A), _ptr(Bt),
                                     _stream_ptr(self.device)), "uce_gram")
        return A, Bt

    def solve_delta(self, A: torch.Tensor, Bt: torch.Tensor) -> torch.Tensor:
        d = A.shape[0]
        DeltaT = torch.empty(d, d, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_solve_delta(self._h, _ptr(A), _ptr(Bt), d, _ptr(DeltaT), _stream_ptr(self.device)),
                   "uce_solve_delta")
        return DeltaT

    def solve_rhs(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """X [d, m] f32 = A^-1 B for an SPD f64 A [d, d] (destroyed) and an f64 B [d, m], m % 64 == 0."""
        d, m = B.shape
        X = torch.empty(d, m, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_solve_rhs(self._h, _ptr(A), _ptr(B), d, m, _ptr(X), _stream_ptr(self.device)), "uce_solve_rhs")
        return X

    def solve_general(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """X [n, m] f32 = A^-1 B for ANY non-singular f64 A [n, n] (destroyed) and f64 B [n, m] (destroyed): Gaussian elimination with
        partial pivoting in f64 (uce_solve_general) - the indefinite systems the Cholesky path cannot take.  Raises UceError(EDOM) for
        a matrix that is singular to working precision."""
        n, m = B.shape
        X = torch.empty(n, m, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_solve_general(self._h, _ptr(A), _ptr(B), n, m, _ptr(X), _stream_ptr(self.device)), "uce_solve_general")
        info = ctypes.c_int(0)
        rc = self.lib.uce_status(self._h, ctypes.byref(info), _stream_ptr(self.device))
        if rc == _lib.EDOM:
            raise _lib.UceError(rc, f"solve of the indefinite system (singular to working precision at column {info.value - 1})")
        _lib.check(rc, "uce_status")
        return X

    def apply(self, W_old: torch.Tensor, DeltaT: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows, d = W_old.shape
        out = torch.empty_like(W_old) if out is None else out
        _lib.check(self.lib.uce_apply(self._h, _ptr(W_old), _ptr(DeltaT), _ptr(out), rows, d,
                                      _stream_ptr(self.device)), "uce_apply")
        return out

    def dual_factors(self, C: torch.Tensor, G: Optional[torch.Tensor], s: torch.Tensor, lamb: float):
        N, d = C.shape
        Ne = 0 if G is None else G.shape[0]
        Dm = torch.empty(max(Ne, 1), d, dtype=torch.float32, device=self.device)
        R = torch.empty(max(Ne, 1), d, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_dual_factors(self._h, _ptr(C), _ptr(G), _ptr(s), N, Ne, d, float(lamb), _ptr(Dm),
                                             _ptr(R), _stream_ptr(self.device)), "uce_dual_factors")
        return Dm[:Ne], R[:Ne]

    def apply_lowrank(self, W_old: torch.Tensor, Dm: torch.Tensor, R: torch.Tensor,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows, d = W_old.shape
        out = torch.empty_like(W_old) if out is None else out
        _lib.check(self.lib.uce_apply_lowrank(self._h, _ptr(W_old), _ptr(Dm), _ptr(R), _ptr(out), rows, d,
                                              Dm.shape[0], _stream_ptr(self.device)), "uce_apply_lowrank")
        return out

    def lowrank_project(self, W_old: torch.Tensor, Dm: torch.Tensor) -> torch.Tensor:
        rows, d = W_old.shape
        ne = Dm.shape[0]
        T = torch.empty(rows, (ne + 63) // 64 * 64, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_lowrank_project(self._h, _ptr(W_old), _ptr(Dm), _ptr(T), rows, d, ne,
                                                _stream_ptr(self.device)), "uce_lowrank_project")
        return T

    def lowrank_update(self, W_old: torch.Tensor, T: torch.Tensor, R: torch.Tensor,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows, d = W_old.shape
        out = torch.empty_like(W_old) if out is None else out
        _lib.check(self.lib.uce_lowrank_update(self._h, _ptr(W_old), _ptr(T), _ptr(R), _ptr(out), rows, d,
                                               R.shape[0], _stream_ptr(self.device)), "uce_lowrank_update")
        return out

    def reserve_rows(self, rows_max: int, n_edit_max: int) -> None:
        _lib.check(self.lib.uce_reserve_rows(self._h, rows_max, n_edit_max), "uce_reserve_rows")

    def delta_from_factors(self, Dm: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
        Ne, d = Dm.shape
        DeltaT = torch.empty(d, d, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_delta_from_factors(self._h, _ptr(Dm), _ptr(R), Ne, d, _ptr(DeltaT),
                                                   _stream_ptr(self.device)), "uce_delta_from_factors")
        return DeltaT

    def edit(self, C: torch.Tensor, G: Optional[torch.Tensor], s: torch.Tensor, lamb: float, W_old: torch.Tensor,
             out: Optional[torch.Tensor] = None, algo: int = _lib.ALGO_AUTO, check: bool = False) -> torch.Tensor:
        N, d = C.shape
        Ne = 0 if G is None else G.shape[0]
        rows = W_old.shape[0]
        out = torch.empty_like(W_old) if out is None else out
        _lib.check(self.lib.uce_edit(self._h, _ptr(C), _ptr(G), _ptr(s), N, Ne, d, float(lamb), _ptr(W_old),
                                     _ptr(out), rows, algo, _stream_ptr(self.device)), "uce_edit")
        if check:
            self.status()
        return out

    def status(self) -> None:
        info = ctypes.c_int(0)
        rc = self.lib.uce_status(self._h, ctypes.byref(info), _stream_ptr(self.device))
        if rc == _lib.EDOM:
            raise _lib.UceError(rc, f"solve (leading minor {info.value} is not positive definite)")
        _lib.check(rc, "uce_status")

    def profile(self, fn, iters: int = 20) -> Dict[str, Tuple[float, int]]:
        """Per-kernel average duration of the launches `fn` makes through this handle: {kernel: (avg ms, launches per
        call)} from HIP events the library records around every launch on the launch stream (uce_profile_begin/_end)."""
        fn()
        torch.cuda.synchronize(self.device)
        _lib.check(self.lib.uce_profile_begin(self._h), "uce_profile_begin")
        for _ in range(iters):
            fn()
        buf = ctypes.create_string_buffer(1 << 16)
        _lib.check(self.lib.uce_profile_end(self._h, _stream_ptr(self.device), buf, len(buf)), "uce_profile_end")
        out: Dict[str, Tuple[float, int]] = {}
        for line in buf.value.decode().splitlines():
            name, ms, n = line.split()
            out[name] = (float(ms) / int(n), int(n) // iters)
        return out

    def debias_targets(self, C_edit: torch.Tensor, C_debias: torch.Tensor, Dsum: torch.Tensor) -> torch.Tensor:
        Ne, d = C_edit.shape
        Nd = C_debias.shape[0]
        G = torch.empty_like(C_edit)
        _lib.check(self.lib.uce_debias_targets(self._h, _ptr(C_edit), _ptr(C_debias), _ptr(Dsum), Ne, Nd, d,
                                               _ptr(G), _stream_ptr(self.device)), "uce_debias_targets")
        return G

    def gather_last_token(self, hidden: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """[B, L, d] text-encoder output (bf16 / f16 / f32, on this GPU) + [B] token indices -> [B, d] fp32 rows."""
        B, L, d = hidden.shape
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16, torch.float32: _lib.DTYPE_F32}[hidden.dtype]
        hidden = hidden.contiguous()
        idx = idx.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty(B, d, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.uce_gather_last_token(self._h, _ptr(hidden), _ptr(idx), _ptr(out), B, L, d, dt,
                                                  _stream_ptr(self.device)), "uce_gather_last_token")
        return out

    def cast_bf16(self, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
        assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
        _lib.check(self.lib.uce_cast_bf16(self._h, _ptr(src), _ptr(dst), src.numel(), _stream_ptr(self.device)),
                   "uce_cast_bf16")
        return dst

    def xattn(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
              scale: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, Lq, Cc = q.shape
        Lk = k.shape[1]
        dh = Cc // heads
        scale = dh ** -0.5 if scale is None else scale
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[q.dtype]
        out = torch.empty_like(q) if out is None else out
        _lib.check(self.lib.uce_xattn_fwd(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Lq, Lk, dh,
                                          float(scale), dt, _stream_ptr(self.device)), "uce_xattn_fwd")
        return out

    def sattn(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
              scale: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Self-attention (any Lk) through uce_sattn_fwd; [B, L, H*dh] in and out."""
        B, Lq, Cc = q.shape
        Lk = k.shape[1]
        dh = Cc // heads
        scale = dh ** -0.5 if scale is None else scale
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[q.dtype]
        out = torch.empty_like(q) if out is None else out
        _lib.check(self.lib.uce_sattn_fwd(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Lq, Lk, dh,
                                          float(scale), dt, _stream_ptr(self.device)), "uce_sattn_fwd")
        return out

    def sattn_packed(self, qkv: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
        """Self-attention on a packed projection qkv [B, L, 3 * C] (q | k | v columns) through uce_sattn_packed_fwd -> [B, L, C]."""
        B, Lq, C3 = qkv.shape
        Cc = C3 // 3
        dh = Cc // heads
        scale = dh ** -0.5 if scale is None else scale
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[qkv.dtype]
        out = torch.empty(B, Lq, Cc, dtype=qkv.dtype, device=qkv.device)
        _lib.check(self.lib.uce_sattn_packed_fwd(self._h, _ptr(qkv), _ptr(out), B, heads, Lq, dh, float(scale), dt,
                                                 _stream_ptr(self.device)), "uce_sattn_packed_fwd")
        return out

    LOG2E = 1.4426950408889634

    def sattn_exp2_form(self, B: int, heads: int, L: int, dh: int) -> bool:
        """True where uce_sattn_packed_exp2_fwd has its own kernel form for the shape (scores leave the matrix pipe as exp2's argument)."""
        return bool(self.lib.uce_sattn_exp2_form(self._h, B, heads, L, dh))

    def linear_colscale(self, x: torch.Tensor, weight: torch.Tensor, scale_cols: int, scale: float) -> torch.Tensor:
        """`x @ weight.T` with columns [0, scale_cols) multiplied by `scale` in the f32 accumulator (uce_linear_colscale_fwd)."""
        K, N = x.shape[-1], weight.shape[0]
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        y = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
        _lib.check(self.lib.uce_linear_colscale_fwd(self._h, _ptr(x2), x2.stride(0), _ptr(w), _ptr(y), N, x2.shape[0], N, K,
                                                    int(scale_cols), float(scale), dt, _stream_ptr(self.device)),
                   "uce_linear_colscale_fwd")
        return y

    def sattn_packed_exp2(self, qkv: torch.Tensor, heads: int) -> torch.Tensor:
        """Self-attention on a packed projection whose q columns carry dh^-0.5 * log2(e) (linear_colscale) -> [B, L, C]."""
        B, Lq, C3 = qkv.shape
        Cc = C3 // 3
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[qkv.dtype]
        out = torch.empty(B, Lq, Cc, dtype=qkv.dtype, device=qkv.device)
        _lib.check(self.lib.uce_sattn_packed_exp2_fwd(self._h, _ptr(qkv), _ptr(out), B, heads, Lq, Cc // heads, dt,
                                                      _stream_ptr(self.device)), "uce_sattn_packed_exp2_fwd")
        return out

    def groupnorm_nhwc(self, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, groups: int, eps: float,
                       silu: bool, addend: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """GroupNorm (+ SiLU) of a channels-last [N, C, H, W] tensor through uce_groupnorm_nhwc_fwd; `addend` [N, C]
        (optional) is added per (sample, channel) before the normalisation.  `x2` (channels-last, same N, H, W): the norm of
        torch.cat([x, x2], dim=1) without the concatenation (uce_groupnorm_cat_nhwc_fwd)."""
        N, C1, Hh, Ww = x.shape
        Cc = C1 + (0 if x2 is None else x2.shape[1])
        hw = Hh * Ww
        y = torch.empty_like(x) if x2 is None else \
            torch.empty((N, Cc, Hh, Ww), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        ws = torch.empty(N * self.lib.uce_groupnorm_chunks(hw) * groups * 2, dtype=torch.float32, device=x.device)
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        ald = 0
        if addend is not None:
            if addend.dim() != 2 or addend.stride(1) != 1 or addend.stride(0) % 8 or addend.data_ptr() % 16:
                addend = addend.contiguous()
            ald = addend.stride(0)
        if x2 is not None:
            _lib.check(self.lib.uce_groupnorm_cat_nhwc_fwd(self._h, _ptr(x), _ptr(x2), C1, _ptr(addend), _ptr(weight), _ptr(bias),
                                                           _ptr(y), _ptr(ws), N, hw, Cc, groups, float(eps), int(silu), dt, ald,
                                                           _stream_ptr(self.device)), "uce_groupnorm_cat_nhwc_fwd")
            return y
        _lib.check(self.lib.uce_groupnorm_nhwc_fwd(self._h, _ptr(x), _ptr(addend), _ptr(weight), _ptr(bias), _ptr(y),
                                                   _ptr(ws), N, hw, Cc, groups, float(eps), int(silu), dt, ald,
                                                   _stream_ptr(self.device)), "uce_groupnorm_nhwc_fwd")
        return y

    def add_bias_nhwc(self, a: torch.Tensor, b: Optional[torch.Tensor], bias: Optional[torch.Tensor]) -> torch.Tensor:
        """a + b + bias[c] for channels-last [N, C, H, W] tensors (b, bias optional) through uce_add_bias_nhwc_fwd."""
        N, Cc, Hh, Ww = a.shape
        y = torch.empty_like(a)
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[a.dtype]
        _lib.check(self.lib.uce_add_bias_nhwc_fwd(self._h, _ptr(a), _ptr(b), _ptr(bias), _ptr(y), N * Hh * Ww, Cc, dt,
                                                  _stream_ptr(self.device)), "uce_add_bias_nhwc_fwd")
        return y

    def conv3x3_igemm(self, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                      upsample: bool = False, stride: int = 1, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """3x3 / pad 1 convolution of a channels-last [N, C, H, W] tensor as ONE implicit-GEMM launch (uce_conv3x3_nhwc_fwd): no
        patch matrix.  Cin % 32 == 0, Cout % 8 == 0, channels-last weight.  stride 2 = Downsample2D; `residual` (channels-last,
        the output's shape) is added in the epilogue; both need Cout % 128 == 0 or Cout % 320 == 0."""
        N, Cc, Hs, Ws = x.shape
        Hh, Ww = (2 * Hs, 2 * Ws) if upsample else ((Hs - 1) // stride + 1, (Ws - 1) // stride + 1)
        if stride == 2 and ((Hs | Ws) & 1):
            raise ValueError("stride-2 convolution of an odd-sized image")
        Cout = weight.shape[0]
        y = torch.empty((N, Cout, Hh, Ww), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        xs, ys = x.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1)            # NHWC views of the same storage
        rs = None
        if residual is not None:
            if residual.shape != y.shape or residual.dtype != y.dtype or not residual.is_contiguous(memory_format=torch.channels_last):
                raise ValueError("residual must be a channels-last tensor of the output's shape and dtype")
            rs = residual.permute(0, 2, 3, 1)
        step = even_chunk(N, max(1, ((1 << 31) - 1) // max(1, Hs * Ws * Cc * x.element_size() + 1)))   # < 2 GB per launch
        for n0 in range(0, N, step):
            nb = min(step, N - n0)
            _lib.check(self.lib.uce_conv3x3_nhwc_fwd(self._h, xs[n0].data_ptr(), _ptr(weight), _ptr(bias), ys[n0].data_ptr(),
                                                     nb, Hh, Ww, Cc, Cout, int(upsample), int(stride),
                                                     None if rs is None else rs[n0].data_ptr(), dt, _stream_ptr(self.device)),
                       "uce_conv3x3_nhwc_fwd")
        return y

    def conv3x3_nhwc(self, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                     max_cols_bytes: int = CONV_COLS_BYTES, upsample: bool = False, stride: int = 1,
                     residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """3x3 / pad 1 convolution of a channels-last [N, C, H, W] tensor: ONE implicit-GEMM launch (uce_conv3x3_nhwc_fwd: stride 2,
        fused upsample and the residual epilogue live there) for every shape that kernel family takes (sd.conv_dispatch), else the
        patch matrix through uce_im2col3x3_nhwc + uce_linear_fwd against the channels-last weight viewed as [Cout, 9*C] (+ the
        residual join by uce_add_bias_nhwc_fwd), the batch walked in evenly sized chunks whose patch matrix stays under
        `max_cols_bytes`.  No library GEMM / convolution on either path.
        upsample: convolve the 2x nearest-neighbour upsampling of x (output [N, Cout, 2H, 2W]) without materialising it."""
        N, Cc, Hs, Ws = x.shape
        Hh, Ww = (2 * Hs, 2 * Ws) if upsample else (Hs // stride, Ws // stride)
        Cout = weight.shape[0]
        if conv_takes_igemm(Cc, Cout, stride, residual is not None) and max_cols_bytes == CONV_COLS_BYTES:
            return self.conv3x3_igemm(x, weight, bias, upsample=upsample, stride=stride, residual=residual)
        if stride != 1:
            raise ValueError(f"stride-2 convolution {Cc} -> {Cout}: the implicit-GEMM kernels need Cin % 64 == 0 (or Cin % 32 == 0 "
                             "with an output that 128 / 320-wide tiles divide)")
        if (9 * Cc) % 32 or Cout % 4:
            raise ValueError(f"3x3 convolution {Cc} -> {Cout}: the patch-matrix path needs 9 * Cin % 32 == 0 and Cout % 4 == 0")
        wmat = weight.permute(0, 2, 3, 1).reshape(Cout, 9 * Cc)      # a view for channels_last weights
        y = torch.empty((N, Cout, Hh, Ww), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        y_rows = y.permute(0, 2, 3, 1).reshape(N * Hh * Ww, Cout)     # NHWC view of the same storage
        per_image = Hh * Ww * 9 * Cc * x.element_size()
        step = even_chunk(N, max_cols_bytes // per_image)
        cols = torch.empty((step * Hh * Ww, 9 * Cc), dtype=x.dtype, device=x.device)
        xs = x.permute(0, 2, 3, 1)                                    # [N, H, W, C] view, contiguous
        for n0 in range(0, N, step):
            nb = min(step, N - n0)
            _lib.check(self.lib.uce_im2col3x3_nhwc(self._h, xs[n0].data_ptr(), _ptr(cols), nb, Hh, Ww, Cc, int(upsample),
                                                   _stream_ptr(self.device)), "uce_im2col3x3_nhwc")
            rows = nb * Hh * Ww
            self.linear(cols[:rows], wmat, bias, out=y_rows[n0 * Hh * Ww:n0 * Hh * Ww + rows])
        if residual is not None:
            return self.add_bias_nhwc(y, residual, None)
        return y

    def conv3x3_c4(self, x: torch.Tensor, wmat: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        """3x3 / pad 1 convolution of a channels-last [N, 4, H, W] tensor (conv_in on the latents): uce_im2col3x3_c4 + one
        uce_linear_fwd against `wmat` [Cout, 64] (sd.unet.conv_c4_weight: the 36 taps x channels, zero-padded)."""
        N, Cc, Hh, Ww = x.shape
        assert Cc == 4 and wmat.shape[1] == 64
        Cout = wmat.shape[0]
        cols = torch.empty((N * Hh * Ww, 64), dtype=x.dtype, device=x.device)
        xs = x.permute(0, 2, 3, 1)
        xs = xs if xs.is_contiguous() else xs.contiguous()
        _lib.check(self.lib.uce_im2col3x3_c4(self._h, _ptr(xs), _ptr(cols), N, Hh, Ww, _stream_ptr(self.device)), "uce_im2col3x3_c4")
        y = torch.empty((N, Cout, Hh, Ww), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        self.linear(cols, wmat, bias, out=y.permute(0, 2, 3, 1).reshape(N * Hh * Ww, Cout))
        return y

    def layernorm(self, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
                  residual: Optional[torch.Tensor] = None):
        """LayerNorm over the last dim of a contiguous 16-bit tensor through uce_layernorm_fwd.  With `residual`,
        returns (x + residual, LN(x + residual)) from one pass; without, LN(x)."""
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        y = torch.empty_like(x)
        s = torch.empty_like(x) if residual is not None else None
        _lib.check(self.lib.uce_layernorm_fwd(self._h, _ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(y), _ptr(s),
                                              rows, Cc, float(eps), dt, _stream_ptr(self.device)), "uce_layernorm_fwd")
        return y if residual is None else (s, y)

    def cfg_pndm_step(self, eps: torch.Tensor, cfg: bool, guidance: float, hist, weights, sample: torch.Tensor, cs: float,
                      ce: float):
        """Guidance combine + PLMS step in one launch (uce_cfg_pndm_step).  eps: [2n, ...] (uncond; cond) when cfg else
        [n, ...]; hist: up to three earlier model outputs (most recent first); weights: 4 floats.  -> (eps_guided, prev)."""
        n = sample.numel()
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[sample.dtype]
        eps_out, prev = torch.empty_like(sample), torch.empty_like(sample)
        hs = [None if i >= len(hist) else hist[i] for i in range(3)]
        w = (ctypes.c_float * 4)(*[float(x) for x in weights])
        _lib.check(self.lib.uce_cfg_pndm_step(self._h, _ptr(eps), int(cfg), float(guidance), _ptr(hs[0]), _ptr(hs[1]), _ptr(hs[2]),
                                              w, _ptr(sample), float(cs), float(ce), _ptr(eps_out), _ptr(prev), n, dt,
                                              _stream_ptr(self.device)), "uce_cfg_pndm_step")
        return eps_out, prev

    def linear(self, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, geglu: bool = False, out: Optional[torch.Tensor] = None,
               x2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`x @ weight.T (+ bias) (+ residual)` over the last dim of x through uce_linear_fwd (16-bit, f32 accumulate); with
        `geglu`, weight / bias are the interleaved rows of a GEGLU projection (sd.unet.geglu_interleave) and the result is
        `hidden * gelu(gate)` with half as many columns.  x / residual / out may be row-strided 2-D views (last dim
        contiguous): slices of wider tensors are read and written in place.  `x2` [..., K2]: the layer applied to
        torch.cat([x, x2], dim=-1) without the concatenation (uce_linear_cat_fwd; weight [N, K + K2])."""
        if x2 is not None:
            if geglu or out is not None:
                raise ValueError("the two-source form has the plain epilogue and allocates its output")
            if x2.shape[:-1] != x.shape[:-1] or x2.dtype != x.dtype:
                raise ValueError("x and x2 must agree in every dimension but the last, and in dtype")
            K1, K2 = x.shape[-1], x2.shape[-1]
            N = weight.shape[0]
            dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
            xa, xb = x.reshape(-1, K1), x2.reshape(-1, K2)
            def aligned(t):
                return t if (t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0) else t.contiguous()
            xa, xb = aligned(xa), aligned(xb)
            M = xa.shape[0]
            y = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
            r2 = None if residual is None else residual.reshape(-1, N)
            if r2 is not None and (r2.stride(1) != 1 or r2.stride(0) % 4 or r2.data_ptr() % 8):
                r2 = r2.contiguous()
            ldr = 0 if r2 is None else r2.stride(0)
            w = weight if weight.is_contiguous() else weight.contiguous()
            _lib.check(self.lib.uce_linear_cat_fwd(self._h, _ptr(xa), xa.stride(0), _ptr(xb), xb.stride(0), K1, _ptr(w), _ptr(bias),
                                                   _ptr(r2), ldr, _ptr(y), N, M, N, K1 + K2, _lib.EPILOGUE_NONE, dt,
                                                   _stream_ptr(self.device)), "uce_linear_cat_fwd")
            return y
        K = x.shape[-1]
        N = weight.shape[0]
        n_out = N // 2 if geglu else N
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]

        def rows2d(t: torch.Tensor, cols: int):
            """(tensor, row stride) of a [..., cols] tensor as M rows: contiguous, or a 2-D view with a contiguous last dim"""
            if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= cols and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0:
                return t, t.stride(0)
            t = t.contiguous()                       # (also: a view at an odd offset / row stride - the kernel moves 16-byte pieces)
            return t, cols

        x2, ldx = rows2d(x, K)
        M = x.numel() // K
        if out is None:
            y = torch.empty(*x.shape[:-1], n_out, dtype=x.dtype, device=x.device)
            ldy = n_out
        else:
            y, ldy = out, (out.stride(0) if out.dim() == 2 else n_out)
        r2, ldr = (None, 0) if residual is None else rows2d(residual, N)
        w = weight if weight.is_contiguous() else weight.contiguous()
        _lib.check(self.lib.uce_linear_fwd(self._h, _ptr(x2), ldx, _ptr(w), _ptr(bias), _ptr(r2), ldr, _ptr(y), ldy, M, N, K,
                                           _lib.EPILOGUE_GEGLU if geglu else _lib.EPILOGUE_NONE, dt, _stream_ptr(self.device)),
                   "uce_linear_fwd")
        return y

    def linear_f32(self, x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """`x @ weight.T` with an f32 result ([M, N]; x [M, K], weight [N, K] 16-bit): attention scores ahead of softmax_rows."""
        M, K = x.shape
        N = weight.shape[0]
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.uce_linear_fwd(self._h, _ptr(x), x.stride(0), _ptr(weight), None, None, 0, _ptr(y), N, M, N, K,
                                           _lib.EPILOGUE_F32, dt, _stream_ptr(self.device)), "uce_linear_fwd")
        return y

    def softmax_rows(self, s: torch.Tensor, scale: float, dtype: torch.dtype) -> torch.Tensor:
        """softmax(scale * s) over the last dim of an f32 [rows, L] tensor -> 16-bit, through uce_softmax_rows."""
        rows, L = s.shape
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[dtype]
        p = torch.empty(rows, L, dtype=dtype, device=s.device)
        _lib.check(self.lib.uce_softmax_rows(self._h, _ptr(s), _ptr(p), rows, L, float(scale), dt, _stream_ptr(self.device)),
                   "uce_softmax_rows")
        return p

    def geglu(self, x: torch.Tensor) -> torch.Tensor:
        """x [..., 2*inner] -> x[..., :inner] * gelu(x[..., inner:]) through uce_geglu_fwd."""
        inner = x.shape[-1] // 2
        rows = x.numel() // x.shape[-1]
        y = torch.empty(*x.shape[:-1], inner, dtype=x.dtype, device=x.device)
        dt = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}[x.dtype]
        _lib.check(self.lib.uce_geglu_fwd(self._h, _ptr(x), _ptr(y), rows, inner, dt, _stream_ptr(self.device)),
                   "uce_geglu_fwd")
        return y


# --------------------------------------------------------------------------------------------
# module discovery + the weight slab  (uce_sd_erase.py:15-22)
# --------------------------------------------------------------------------------------------

def is_uce_module(name: str) -> bool:
    """The reference's name predicate (uce_sd_erase.py:18)."""
    return "attn2" in name and (name.endswith("to_v") or name.endswith("to_k"))


def collect_uce_modules(unet: torch.nn.Module) -> List[Tuple[str, torch.nn.Module]]:
    return [(n, m) for n, m in unet.named_modules() if is_uce_module(n)]


@dataclass
class WeightSlab:
    """Every edited projection's weight, concatenated along rows: [sum(o_m), d] fp32."""
    names: List[str]
    offsets: List[int]      # row offset of each module
    rows: List[int]
    data: torch.Tensor      # [sum rows, d]

    @classmethod
    def from_modules(cls, modules: Sequence[Tuple[str, torch.nn.Module]], device) -> "WeightSlab":
        if not modules:
            raise ValueError("no attn2.to_k / attn2.to_v modules found in the U-Net")
        d = modules[0][1].weight.shape[1]
        rows = [int(m.weight.shape[0]) for _, m in modules]
        offs = [0]
        for r in rows[:-1]:
            offs.append(offs[-1] + r)
        data = torch.empty(sum(rows), d, dtype=torch.float32, device=device)
        for (n, m), o, r in zip(modules, offs, rows):
            if m.weight.shape[1] != d:
                raise ValueError(f"{n}: in_features {m.weight.shape[1]} != {d}")
            data[o:o + r].copy_(m.weight.detach())
        return cls([n for n, _ in modules], offs, rows, data)

    def like(self, data: torch.Tensor) -> "WeightSlab":
        return WeightSlab(self.names, self.offsets, self.rows, data)

    def views(self) -> List[torch.Tensor]:
        return [self.data[o:o + r] for o, r in zip(self.offsets, self.rows)]

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Keys `<module path>.weight` like uce_sd_erase.py:85-87."""
        return {n + ".weight": v for n, v in zip(self.names, self.views())}


def save_uce_state(slab: WeightSlab, save_dir: str, exp_name: str) -> str:
    """safetensors artifact `{save_dir}/{exp_name}.safetensors` (uce_sd_erase.py:85-88): keys `<module path>.weight`, fp32, the
    module's [o, d] shape - what `load_file` + `load_state_dict(strict=False)` of generate-images-sd.py:17-19 consume.

    The slab IS the file's data section (modules in row order, rows contiguous): ONE device -> host copy of the whole slab and one
    write behind a hand-built header, instead of a copy per module, a dict of host tensors and the serializer's own copy of all of
    them (safetensors format: 8-byte little-endian header length, JSON {name: {dtype, shape, data_offsets}}, raw little-endian
    data).  A slab whose modules are not packed back to back in order takes safetensors' save_file."""
    import json as _json
    path = os.path.join(save_dir, exp_name + ".safetensors")
    d = int(slab.data.shape[1])
    packed = (slab.data.dtype == torch.float32 and slab.data.is_contiguous() and len(set(slab.names)) == len(slab.names)
              and all(o == sum(slab.rows[:i]) for i, o in enumerate(slab.offsets)) and sum(slab.rows) == slab.data.shape[0])
    if not packed:
        from safetensors.torch import save_file
        save_file({k: v.detach().cpu().contiguous() for k, v in slab.state_dict().items()}, path)
        return path
    header = {}
    for n, o, r in zip(slab.names, slab.offsets, slab.rows):
        header[n + ".weight"] = {"dtype": "F32", "shape": [int(r), d], "data_offsets": [int(o) * d * 4, int(o + r) * d * 4]}
    hb = _json.dumps(header, separators=(",", ":")).encode("utf-8")
    hb += b" " * ((8 - len(hb) % 8) % 8)                               # (the data section starts 8-byte aligned, as save_file pads)
    with open(path, "wb", buffering=0) as f:
        f.write(len(hb).to_bytes(8, "little") + hb)
        if slab.data.device.type == "cuda":
            _stream_to_file(f, slab.data.detach().view(-1))
        else:
            f.write(memoryview(slab.data.detach().numpy()).cast("B"))
    return path


SAVE_CHUNK_BYTES = 8 << 20
_save_pinned: Dict[int, list] = {}


def _stream_to_file(f, flat: torch.Tensor) -> None:
    """Device tensor -> file through two pinned 8 MB buffers: the copy of chunk i + 1 (its own stream, behind whatever the current
    stream has queued - the edit) runs under the write of chunk i.  The file system is the floor (76.7 MB: ~15 ms on the build's
    boxes); a pageable `.to("cpu")` of the whole slab first cost another 9.7 ms, pinning the whole slab 5.9 ms
    (tools/probe_save.py).  The bytes are those of one write of the whole tensor."""
    total = flat.numel()
    esz = flat.element_size()
    n = max(1, min(SAVE_CHUNK_BYTES // esz, total))
    key = flat.device.index or 0
    bufs = _save_pinned.get(key)
    if bufs is None or bufs[0].numel() * bufs[0].element_size() < n * esz:
        bufs = _save_pinned[key] = [torch.empty(n * esz, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    views = [b[:n * esz].view(flat.dtype) for b in bufs]
    st = torch.cuda.Stream(device=flat.device)
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    nch = (total + n - 1) // n

    def fetch(i):
        lo, hi = i * n, min((i + 1) * n, total)
        with torch.cuda.stream(st):
            views[i & 1][:hi - lo].copy_(flat[lo:hi], non_blocking=True)
            evs[i & 1].record(st)

    st.wait_stream(torch.cuda.current_stream(flat.device))
    fetch(0)
    for i in range(nch):
        evs[i & 1].synchronize()
        if i + 1 < nch:
            fetch(i + 1)            # into the other buffer: its previous chunk (i - 1) is on its way to the file already
        lo, hi = i * n, min((i + 1) * n, total)
        f.write(memoryview(bufs[i & 1].numpy())[:(hi - lo) * esz])
    torch.cuda.current_stream(flat.device).wait_stream(st)


# --------------------------------------------------------------------------------------------
# concept embeddings  (uce_sd_erase.py:25-42)
# --------------------------------------------------------------------------------------------

AUTO_EMBED_BATCH = 64
AUTO_EMBED_MAX = 1024        # strings per forward at most (automatic mode: see last_token_embeddings)


def last_token_embeddings(pipe, prompts: Sequence[str], device, cache: Optional[Dict[str, torch.Tensor]] = None,
                          batch_size: Optional[int] = 0) -> Dict[str, torch.Tensor]:
    """Per UNIQUE string: text-encoder hidden state at index `attention_mask.sum() - 2`
    (the last real token; '' -> the BOS position).  Returns {prompt: [d] fp32 on `device`}.

    batch_size == 0 reproduces the reference call pattern exactly (one `encode_prompt` per unique
    string, uce_sd_erase.py:26-42).  batch_size > 0 is SURVEY.md section 8(f) row 1: all unique
    strings go through the text encoder in batches of that size and the last-token rows are
    gathered on the device - the text encoder, not the closed-form solve, is what a 1 500-concept
    edit spends its wall-clock on.  batch_size None / < 0: automatic (64 for the build's own pipeline on a GPU, else per string)."""
    out = {} if cache is None else cache
    todo = list(dict.fromkeys(e for e in prompts if e not in out))     # unique, first-seen order
    auto = batch_size is None or batch_size < 0
    if auto:
        # automatic: the build's own pipeline on a GPU batches (the rows of a batch are independent and the gather is bit-exact,
        # tests/test_host_cpu.py, test_edit_gpu.py); a foreign pipe object keeps the reference's call pattern
        from .sd import pipeline as _sdp
        own = isinstance(pipe, _sdp.StableDiffusionPipeline) and torch.device(device).type == "cuda"
        batch_size = AUTO_EMBED_BATCH if own else 0
    if batch_size and batch_size > 1 and len(todo) > 1:
        prefix_ok = hasattr(pipe, "encode_prompt_prefix")
        T = pipe.tokenizer.model_max_length
        tok_all = pipe.tokenizer(todo, padding="max_length", max_length=T, truncation=True, return_tensors="pt")     # once: the ids
        idx_all = tok_all["attention_mask"].sum(dim=1) - 2                                                          # go down with it
        i = 0
        while i < len(todo):
            B = batch_size
            if auto and prefix_ok:
                # the automatic batch is a budget of TOKEN POSITIONS (64 strings x 77 positions), not of strings: with the encoder cut
                # to the positions a batch needs (below), short concept names go through in batches of up to 1024 - the 200
                # launches of a forward cost the same for 64 x 5 positions as for 1024 x 5
                B = min(AUTO_EMBED_MAX, len(todo) - i)
                while B > batch_size and B * (max(int(idx_all[i:i + B].max()), 0) + 1) > batch_size * T:
                    B = max(batch_size, B // 2)
            chunk, idx = todo[i:i + B], idx_all[i:i + B]
            i += len(chunk)
            if prefix_ok and int(idx.min()) >= 0:
                # the build's own pipeline: the causal encoder on positions 0 .. max(idx) only (a concept name ends at position
                # 2-8 of 77; the states of a prefix do not depend on what follows it)
                t_emb = pipe.encode_prompt_prefix(chunk, device, int(idx.max()) + 1,
                                                  input_ids=tok_all["input_ids"][i - len(chunk):i])     # [B, max(idx) + 1, d]
            else:
                t_emb = pipe.encode_prompt(prompt=chunk, device=device, num_images_per_prompt=1,
                                           do_classifier_free_guidance=False)[0]        # [B, 77, d]
            idx = torch.where(idx < 0, idx + t_emb.shape[1], idx)      # python indexing of the reference: -1 wraps
            if t_emb.is_cuda and t_emb.shape[-1] % 8 == 0:
                rows = UceHandle.get(t_emb.device).gather_last_token(t_emb, idx)     # fused gather + fp32 widening (HIP)
            else:
                rows = t_emb[torch.arange(len(chunk), device=t_emb.device), idx.to(t_emb.device), :]
            for e, r in zip(chunk, rows):
                out[e] = r.to(device=device, dtype=torch.float32)
        return out
    for e in todo:
        t_emb = pipe.encode_prompt(prompt=e, device=device, num_images_per_prompt=1,
                                   do_classifier_free_guidance=False)
        mask = pipe.tokenizer(e, padding="max_length", max_length=pipe.tokenizer.model_max_length,
                              truncation=True, return_tensors="pt")["attention_mask"]
        idx = int(mask.sum()) - 2
        out[e] = t_emb[0][0, idx, :].to(device=device, dtype=torch.float32)
    return out


def concept_matrices(embeds: Dict[str, torch.Tensor], edit: Sequence[str], guide: Sequence[str],
                     preserve: Sequence[str], erase_scale: float, preserve_scale: float, device
                     ) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """C [N,d] (edit rows in list order, then preserve rows), G [N_e,d], s [N].  The sums of
    the reference iterate over the LISTS (uce_sd_erase.py:66,74), so duplicates stay."""
    if len(edit) != len(guide):
        raise ValueError("edit and guide concept lists differ in length")
    rows = [embeds[e] for e in edit] + [embeds[p] for p in preserve]
    C = torch.stack(rows).to(device=device, dtype=torch.float32).contiguous()
    G = torch.stack([embeds[g] for g in guide]).to(device=device, dtype=torch.float32).contiguous() if edit else None
    s = torch.tensor([float(erase_scale)] * len(edit) + [float(preserve_scale)] * len(preserve),
                     dtype=torch.float32, device=device)
    return C, G, s


def drop_zero_scale_rows(C, G, s, n_edit: int):
    """Rows with scale 0 contribute nothing to either sum; remove them (the dual form divides by s)."""
    keep = (s != 0)
    if bool(keep.all()):
        return C, G, s, n_edit
    ke = keep[:n_edit]
    G2 = G[ke].contiguous() if G is not None else None
    return C[keep].contiguous(), G2, s[keep].contiguous(), int(ke.sum())


# --------------------------------------------------------------------------------------------
# the drop-in entry points
# --------------------------------------------------------------------------------------------

def check_spd_inputs(scales, lamb: float) -> bool:
    """True when the closed-form system lamb*I + sum_i s_i c_i c_i^T is symmetric POSITIVE DEFINITE by construction (all
    scales >= 0, lamb > 0): the Cholesky solver's case.  Host-side look at the few scalars a job is built from - callers
    that build `s` themselves run it ONCE (no device read-back per edit; the device-side pivot check reports through
    uce_status).  False: negative scales / lamb <= 0, which the reference's general LU inverse (torch.inverse,
    uce_sd_erase.py:82) accepts - edit_slab then takes the normal-equations form below."""
    vals = [float(v) for v in (scales.tolist() if isinstance(scales, torch.Tensor) else scales)]
    return not (any(v < 0 for v in vals) or not (lamb > 0))


def edit_slab_general(handle: UceHandle, slab: WeightSlab, C: torch.Tensor, G: torch.Tensor, s: torch.Tensor,
                      lamb: float) -> WeightSlab:
    """The edit for a symmetric INDEFINITE system (negative scales or lamb <= 0; reference: the same
    `mat1 @ torch.inverse(mat2)` - its LU inverse does not care about definiteness).  An EDGE path outside the SPD hot path, on the
    library's own kernels end to end: the f64 Gram of the primal form (uce_gram), the d x d solve Delta^T = A^-1 Bt by Gaussian
    elimination with partial pivoting in f64 (uce_solve_general - no vendor solver), the dense apply (uce_apply).  Error
    ~1e-16 d cond(A), below the reference's fp32 LU inverse for every A that one can invert at all; a singular A raises like the
    reference's torch.inverse does."""
    A, Bt = handle.gram(C, G, s, lamb)
    DT = handle.solve_general(A, Bt)
    return slab.like(handle.apply(slab.data, DT))


def edit_slab(handle: UceHandle, slab: WeightSlab, C: torch.Tensor, G: Optional[torch.Tensor], s: torch.Tensor,
              lamb: float, algo: int = _lib.ALGO_AUTO, validated: bool = False, spd: bool = True) -> WeightSlab:
    """`validated`: the caller has run check_spd_inputs / drop_zero_scale_rows itself and passes `spd`."""
    n_edit = 0 if G is None else G.shape[0]
    if not validated:
        spd = check_spd_inputs(s, lamb)
        C, G, s, n_edit = drop_zero_scale_rows(C, G, s, n_edit)
    if C.shape[0] == 0 or n_edit == 0:
        # nothing pulls the weights anywhere: W_new = W_old exactly (Delta = 0)
        return slab.like(slab.data.clone())
    if G is not None and G.shape[0] == 0:
        G = None
    if not spd:
        return edit_slab_general(handle, slab, C, G, s, lamb)
    out = handle.edit(C, G, s, lamb, slab.data, algo=algo, check=True)
    return slab.like(out)


def UCE(pipe, edit_concepts, guide_concepts, preserve_concepts, erase_scale, preserve_scale, lamb, save_dir,
        exp_name, device: str = "cuda:0", algo: int = _lib.ALGO_AUTO, return_slab: bool = False,
        embed_batch: Optional[int] = None, timings: Optional[Dict[str, float]] = None):
    """Same positional signature and artifact as the reference's UCE() (uce_sd_erase.py:12);
    `device` replaces the module global the reference reads.  `timings` (optional dict) receives the wall seconds of the
    stages - slab (module discovery + packing), embed (text encoder, uce_sd_erase.py:25-42), edit (:45-82, device-synchronised),
    save (:85-88: device -> host + safetensors) - that add up to the reference's own "Model edited in X seconds"."""
    start_time = time.time()
    dev = torch.device(device)
    handle = UceHandle.get(dev)

    def lap(name, t_prev):
        if timings is None:
            return t_prev
        torch.cuda.synchronize(handle.device)
        now = time.time()
        timings[name] = timings.get(name, 0.0) + (now - t_prev)
        return now

    t = start_time
    modules = collect_uce_modules(pipe.unet)
    slab = WeightSlab.from_modules(modules, handle.device)
    t = lap("slab", t)
    embeds = last_token_embeddings(pipe, list(edit_concepts) + list(guide_concepts) + list(preserve_concepts),
                                   handle.device, batch_size=embed_batch)
    t = lap("embed", t)
    C, G, s = concept_matrices(embeds, edit_concepts, guide_concepts, preserve_concepts, erase_scale,
                               preserve_scale, handle.device)
    new = edit_slab(handle, slab, C, G, s, lamb, algo)
    t = lap("edit", t)
    path = save_uce_state(new, save_dir, exp_name)
    t = lap("save", t)
    end_time = time.time()
    if timings is not None:
        timings["total"] = end_time - start_time
    print(f"\n\nErased concepts using UCE\nModel edited in {end_time - start_time} seconds\n")
    return (new, path) if return_slab else None


def debias_keys_alias(edit_keys: Sequence, debias_keys: Sequence, preserve_keys: Sequence) -> bool:
    """True when the reference's string-keyed cache (uce_sd_debias.py:69-88) makes two rows share ONE drifting tensor: an
    edit concept listed twice, or an edit concept that is also a debias or a preserve concept (:122-127 mutate the cached
    guide output of the EDIT concept in place; debias-only and preserve-only strings are never written)."""
    e = list(edit_keys)
    return len(set(e)) != len(e) or bool(set(e) & (set(debias_keys) | set(preserve_keys)))


def debias_alias_step(keys, uniq: Sequence, coef: Dict, D: np.ndarray):
    """One iteration of uce_sd_debias.py:120-133 on the coefficients of the cached guide outputs (`coef[k]`: float64 row over
    `uniq`, updated IN PLACE like the reference's tensors; g_k = coef[k] @ C_uniq).  Returns
      Dm         [n_e + len(moved_pres), len(uniq)]  target minus the row's own embedding, as coefficients
      moved_pres preserve rows whose cached output has drifted (they are edit rows of this iteration's system)
      pure       preserve rows still equal to their own embedding."""
    ek, dk, pk = keys
    pos = {k: i for i, k in enumerate(uniq)}
    moved, own = [], []
    for idx, e in enumerate(ek):                                 # :120-127, in list order; right-hand side first
        for i, concept in enumerate(dk):
            coef[e] = coef[e] + float(D[idx][i]) * coef[concept]
        moved.append(coef[e].copy())
        own.append(pos[e])
    moved_pres, pure = [], []
    for j, p_ in enumerate(pk):                                  # :132-133: the cached output as it is NOW
        c = coef[p_]
        if np.count_nonzero(c) == 1 and c[pos[p_]] == 1.0:
            pure.append(j)
        else:
            moved.append(c.copy())
            own.append(pos[p_])
            moved_pres.append(j)
    Dm = np.stack(moved)
    Dm[np.arange(len(own)), own] -= 1.0                          # uce_debias_targets adds the row's own embedding
    return Dm, moved_pres, pure


class DebiasState:
    """Iteration state of the debias loop (uce_sd_debias.py:95-141) in closed form: the drift
    the reference adds in place to its cached guide outputs accumulates, and every iteration
    re-solves from W_old, so the weights after iteration t depend only on sum_{t'<=t} D_t'.

    `keys` = (edit, debias, preserve) string lists.  The reference caches ONE guide output per unique string, so when the
    lists alias (debias_keys_alias) the drift of one row is seen by another: the state then follows the reference's update
    order on the COEFFICIENTS of every cached output over the unique embeddings (the outputs are W g_x with g_x a linear
    combination of embeddings - module independent), a few scalars on the host per iteration, and hands the targets to
    the same device path (uce_debias_targets + uce_edit)."""

    def __init__(self, handle: UceHandle, slab: WeightSlab, C_edit: torch.Tensor, C_debias: torch.Tensor,
                 C_pres: Optional[torch.Tensor], edit_scale: float, preserve_scale: float, lamb: float,
                 algo: int = _lib.ALGO_AUTO, keys: Optional[Tuple[Sequence, Sequence, Sequence]] = None):
        self.handle, self.slab, self.lamb, self.algo = handle, slab, lamb, algo
        self.C_edit, self.C_debias = C_edit, C_debias
        n_e = C_edit.shape[0]
        n_p = 0 if C_pres is None else C_pres.shape[0]
        self.current = slab.like(slab.data.clone())
        self.aliased = keys is not None and debias_keys_alias(*keys)
        if self.aliased:
            ek, dk, pk = (list(k) for k in keys)
            if (len(ek), len(dk), len(pk)) != (n_e, C_debias.shape[0], n_p):
                raise ValueError("debias keys do not match the embedding rows")
            self.keys = (ek, dk, pk)
            uniq: List = []
            rows = []
            for k, r in zip(ek + dk + pk, list(C_edit) + list(C_debias) + (list(C_pres) if n_p else [])):
                if k not in uniq:
                    uniq.append(k)
                    rows.append(r)
            self.uniq = uniq
            self.C_uniq = torch.stack(rows).contiguous()
            self.coef = {k: np.eye(len(uniq), dtype=np.float64)[i] for i, k in enumerate(uniq)}   # g_k = coef[k] @ C_uniq
            self.C_pres = C_pres
            self.scales = (float(edit_scale), float(preserve_scale))
            return
        if float(preserve_scale) == 0.0:                         # zero-scale rows contribute nothing: dropped here, once
            n_p = 0
        self.no_edit = float(edit_scale) == 0.0                  # nothing pulls the weights anywhere
        self.C = torch.cat([C_edit] + ([C_pres] if n_p else [])).contiguous()
        self.s = torch.tensor([float(edit_scale)] * n_e + [float(preserve_scale)] * n_p, dtype=torch.float32,
                              device=handle.device)
        self.spd = check_spd_inputs([edit_scale, preserve_scale], lamb)   # once: the per-iteration edits skip the read-back of `s`
        self.Dsum = torch.zeros(n_e, C_debias.shape[0], dtype=torch.float64, device=handle.device)

    def _step_aliased(self, D: np.ndarray) -> WeightSlab:
        Dm, moved_pres, pure = debias_alias_step(self.keys, self.uniq, self.coef, D)
        n_e = len(self.keys[0])
        Cf = torch.cat([self.C_edit] + [self.C_pres[j:j + 1] for j in moved_pres]).contiguous()
        G = self.handle.debias_targets(Cf, self.C_uniq, torch.as_tensor(Dm, device=self.handle.device))
        C = torch.cat([Cf] + [self.C_pres[j:j + 1] for j in pure]).contiguous()
        s = torch.tensor([self.scales[0]] * n_e + [self.scales[1]] * (len(moved_pres) + len(pure)), dtype=torch.float32,
                         device=self.handle.device)
        # rows with scale 0 still drift (above) but contribute nothing: edit_slab drops them
        self.current = edit_slab(self.handle, self.slab, C, G, s, self.lamb, self.algo)
        return self.current

    def step(self, direction_scale: np.ndarray) -> WeightSlab:
        if self.aliased:
            return self._step_aliased(np.asarray(direction_scale, dtype=np.float64))
        self.Dsum += torch.as_tensor(np.asarray(direction_scale, dtype=np.float64), device=self.handle.device)
        if self.no_edit:
            return self.current
        G = self.handle.debias_targets(self.C_edit, self.C_debias, self.Dsum)
        self.current = edit_slab(self.handle, self.slab, self.C, G, self.s, self.lamb, self.algo, validated=True, spd=self.spd)
        return self.current
