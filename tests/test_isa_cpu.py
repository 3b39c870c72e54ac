"""The memory schedule of the whole-row epilogues, read back from the ISA hipcc emits (no GPU needed: hipcc cross-compiles).

DESIGN 4.38: a gfx9 wave has ONE vmcnt for loads and stores and they complete in order, so an epilogue that waits `vmcnt(0)` in
front of every residual piece waits for the acknowledgement of the store before it.  The rewritten epilogues are straight-line
buffer-operation code whose waits are exact counts; a compiler or source change that brings the per-store `vmcnt(0)` (or the
conditional blocks the loads were sunk into) back would cost the 3.7 % it bought without failing any numerics test - this does."""
import os
import re
import shutil
import subprocess

import pytest

from uce_amd import build as B

CSRC = os.path.join(os.path.dirname(B.__file__), "csrc")


_ASM = {}


def _device_asm(tmp_path_factory, source: str) -> str:
    if source not in _ASM:                                               # (one compilation per source file and session)
        cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(cc):
            pytest.skip("hipcc not installed")
        out = str(tmp_path_factory.mktemp("isa") / (source + ".s"))
        subprocess.run([cc] + list(B.FLAGS) + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, source)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _ASM[source] = open(out).read()
    return _ASM[source]


def _kernel(asm: str, mangled_part: str) -> list:
    m = re.search(r"^(_ZN\S*" + re.escape(mangled_part) + r"\S*):", asm, re.M)
    assert m, f"kernel {mangled_part} not in the object"
    body = asm[m.start():]
    return body[:body.index("s_endpgm")].split("\n")


def _epilogue(lines: list) -> list:
    """Everything behind the main loop in the listing: after the last MFMA AND the last LDS-DMA (block placement may put the
    loop's staging blocks - with their wave-uniform two-source branch - behind the last MFMA in the text)."""
    last = max(i for i, l in enumerate(lines) if "v_mfma" in l or (" lds" in l and "buffer_load" in l))
    return lines[last:]


@pytest.mark.parametrize("source,kernel,min_stores", [
    ("uce_gemm.hip", "k_gemm_dmaILi4ELi2ELi2ELi5ELb0ELb0ELb1ELi2ELi64ELb0ELi8E", 40),     # 256 x 320 tiles, bias (+ residual): both paths
    ("uce_gemm.hip", "k_gemm_dmaILi2ELi4ELi4ELi2ELb0ELb1ELb1ELi2ELi64ELb0ELi8E", 8),      # 256 x 256 tiles, GEGLU
    ("uce_conv_w1.hip", "k_conv3x3_w1ILi10ELb0ELi9E", 80),                                 # one wave per SIMD, 256 x 320 tiles
])
def test_whole_row_epilogue_is_straight_line_buffer_code_with_counted_waits(tmp_path_factory, source, kernel, min_stores):
    ep = _epilogue(_kernel(_device_asm(tmp_path_factory, source), kernel))
    stores = sum("buffer_store_dwordx4" in l for l in ep)
    assert stores >= min_stores, stores
    # every output / residual access is a buffer operation on the wave's own descriptor
    assert not any(("global_store" in l or "global_load" in l or "flat_store" in l or "flat_load" in l) for l in ep)
    # no divergent block around a load or a store (uniform branches - residual or not - remain)
    assert sum("s_cbranch_exec" in l for l in ep) <= 1
    # counted waits: a handful of full drains per path (the end of the bias batch), not one per store
    drains = sum(bool(re.search(r"s_waitcnt vmcnt\(0\)", l)) for l in ep)
    assert drains <= max(2, stores // 8), (drains, stores)
    assert any(re.search(r"s_waitcnt vmcnt\(([2-9]|1[0-9])\)", l) for l in ep)              # loads awaited past younger operations
