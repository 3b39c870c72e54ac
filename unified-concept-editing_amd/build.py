"""Builds libuce_hip.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
gpurun snapshot.  `python -m uce_amd.build` rebuilds unconditionally.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libuce_hip.so")
SOURCES = ["uce_gram.hip", "uce_solve.hip", "uce_apply.hip", "uce_apply_b3.hip", "uce_lowrank2.hip", "uce_xattn.hip", "uce_sattn.hip", "uce_norm.hip", "uce_conv.hip", "uce_api.hip"]
HEADERS = ["uce_common.h", os.path.join("..", "..", "include", "uce_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs (gfx950 reads/writes them there directly).  Left to
    # its heuristics the compiler parks them in AGPRs and pays a v_accvgpr_read/write pair around every VALU
    # touch of an accumulator - 128 extra moves per key tile in the attention kernels' softmax.
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print("[uce_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv or True)
    print(LIB_PATH)
