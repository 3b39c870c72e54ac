"""Which path a 3x3 / pad 1 convolution of the U-Net / VAE takes (sd/unet.py -> UceHandle.conv3x3_nhwc): the implicit-GEMM kernels
(uce_conv3x3_nhwc_fwd) for every shape they exist for, the patch matrix + uce_linear_fwd for the rest, and how a batch is walked
in chunks.  Host logic only; UCE_CONV_COLS_MB overrides the chunk size for measurements (read once, at import)."""
from __future__ import annotations

import os

# patch-matrix chunk of the im2col + GEMM convolutions (bytes); UCE_CONV_COLS_MB overrides for measurements
CONV_COLS_BYTES = int(os.environ.get("UCE_CONV_COLS_MB", "4096")) << 20


def conv_takes_igemm(Cin: int, Cout: int, stride: int = 1, residual: bool = False) -> bool:
    """uce_conv3x3_nhwc_fwd has a kernel for the layer (csrc/uce_conv_igemm.hip's dispatch, restated): 128-byte k-tiles
    (Cin % 64 == 0) on any Cout % 8 == 0 - the wide direct-to-LDS forms where a 128 / 256 / 320-wide tile divides Cout, the few-tile
    forms with the split contraction elsewhere - or 64-byte k-tiles (Cin % 32 == 0) on outputs that a 128- / 320-wide tile
    divides.  Everything else (no layer of SD-1.x / SD-2.x / SDXL or their VAEs) goes through the patch matrix
    (uce_im2col3x3_nhwc) + uce_linear_fwd, which has no stride-2 form.  No library convolution or GEMM is behind either path."""
    if Cin % 32 or Cout % 8:
        return False
    if Cout % 128 == 0 or Cout % 320 == 0:
        return True
    return Cin % 64 == 0


def even_chunk(n: int, cap: int) -> int:
    """Chunk length for walking `n` items at most `cap` at a time: the fewest chunks that respect the cap, evenly
    sized (32 items, cap 15 -> 11 + 11 + 10 rather than 15 + 15 + 2; a small tail launch cannot fill the chip)."""
    cap = max(1, min(n, cap))
    chunks = -(-n // cap)
    return -(-n // chunks)
