"""Which path a 3x3 / stride 1 / pad 1 convolution of the U-Net / VAE takes (sd/unet.py -> UceHandle.conv3x3_nhwc): the
implicit-GEMM kernels (uce_conv3x3_nhwc_fwd) or im2col + one library GEMM, and how a batch is walked in chunks.  Host
logic only - measured on an MI355X with tools/probe_igemm.py; UCE_CONV_IGEMM / UCE_CONV_COLS_MB override it for A/B runs
(read once, at import)."""
from __future__ import annotations

import os

# patch-matrix chunk of the im2col + GEMM convolutions (bytes); UCE_CONV_COLS_MB overrides for measurements
CONV_COLS_BYTES = int(os.environ.get("UCE_CONV_COLS_MB", "4096")) << 20


# which 3x3 convolutions go through the implicit-GEMM kernel instead of im2col + library GEMM: "auto" = where it measured
# faster on an MI355X (tools/probe_igemm.py: the high-resolution, narrow layers of the VAE decoder - 2.7x at 128 -> 128
# channels on 512 x 512 - where the patch matrix is all traffic and no arithmetic), "always", "never"
CONV_IGEMM = os.environ.get("UCE_CONV_IGEMM", "auto")


def conv_prefers_igemm(H: int, W: int, Cin: int, Cout: int, N: int = 32, stride: int = 1) -> bool:
    """Measured on an MI355X (tools/probe_igemm.py, tools/probe_r04.py, bf16): the implicit-GEMM kernels run 800-1110 TF/s
    (direct-to-LDS form, outputs of 128 / 256 / 320-multiples) or 600-820 TF/s (128 x 128 register-staged form) once there are
    enough pixel tiles; the library GEMM (stream-K) reaches 0.85-1.25 PF/s on the small-spatial, wide layers but pays the
    patch-matrix round trip everywhere.  (H, W) = OUTPUT size.  stride 2 (Downsample2D): the direct-to-LDS kernel against
    MIOpen - 81-139 us against 167-191 at the generation batch, wherever there are at least 64 pixel tiles."""
    if CONV_IGEMM == "never" or Cin % 32 or Cout % 8:
        return False
    if CONV_IGEMM == "always":
        return True
    M = N * H * W
    if stride == 2:
        return (Cout % 128 == 0 or Cout % 320 == 0) and M >= 8192
    if Cin % 64 and Cout % 128 and Cout % 320:       # (the register-staged fallback kernel needs 64-channel chunks)
        return False
    if M >= 128 * 1024:                              # U-Net 64 x 64 at the generation batch, every VAE layer >= 128^2
        return True
    # 32 x 32 layers: the direct-to-LDS form (outputs that are multiples of 256 / 320 channels: 835-1113 TF/s against
    # 736-761 for im2col + GEMM) or a long contraction on the 128 x 128 kernel; 16 x 16 and 8 x 8 layers have too few
    # pixel tiles for either (1280 -> 1280 @ 16 x 16 x 32: 284 us with 128-pixel tiles against 225 for im2col + library GEMM;
    # @ 8 x 8: 141 against 93)
    if M >= 32 * 1024 and (Cin >= 640 or Cout % 256 == 0 or Cout % 320 == 0):
        return True
    # 16 x 16 at 32 samples: with 128-byte k-tiles and 128-pixel tiles the direct-to-LDS kernel ties im2col + library GEMM (1280 ->
    # 1280: 229 us against 221) as soon as every CU gets a 128 x 320 tile - and needs no patch matrix
    return Cin % 64 == 0 and Cout % 320 == 0 and -(-M // 128) * (Cout // 320) >= 256


def even_chunk(n: int, cap: int) -> int:
    """Chunk length for walking `n` items at most `cap` at a time: the fewest chunks that respect the cap, evenly
    sized (32 items, cap 15 -> 11 + 11 + 10 rather than 15 + 15 + 2; a small tail launch cannot fill the chip)."""
    cap = max(1, min(n, cap))
    chunks = -(-n // cap)
    return -(-n // chunks)
