"""Builds libuce_hip.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
gpurun snapshot.  Every csrc/*.hip is compiled to its own object (in parallel, only when it or a
header is newer than the object) and the objects are linked into lib/libuce_hip.so.
`python -m uce_amd.build` rebuilds unconditionally.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
PUBLIC_HEADER = os.path.normpath(os.path.join(PKG, "..", "include", "uce_hip.h"))

# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs (gfx950 reads/writes them there directly).  Left to
# its heuristics the compiler parks them in AGPRs and pays a v_accvgpr_read/write pair around every VALU
# touch of an accumulator - 128 extra moves per key tile in the attention kernels' softmax.
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS = list(BASE_FLAGS)
if os.environ.get("UCE_CHAIN_DEBUG"):          # phase stamps of the rider chain (tools/dbg_chain.py); never in the product build
    FLAGS.append("-DUCE_CHAIN_DEBUG")
FLAGS += os.environ.get("UCE_DEFINES", "").split()            # experiment switches (-DNAME=VALUE ...), empty in the product build
# the host-side AddressSanitizer build (tests/test_abi_cpu.py, SURVEY.md section 5 "sanitizers"): the handle's workspace
# management, argument checks and dispatch compiled with ASAN, the device code left alone (gfx950 without xnack has no device ASAN)
ASAN_DEFINES = "-fsanitize=address -fno-gpu-sanitize -shared-libsan -g"
LINK_FLAGS = [f for f in FLAGS if f.startswith("-fsanitize") or f in ("-shared-libsan", "-fno-gpu-sanitize")]

# Objects are only as fresh as the FLAGS they were compiled with: every flag set gets its own object directory, and a
# build with debug / experiment flags links its own library file - the product libuce_hip.so is only ever made of
# product objects (an mtime check alone would re-link a stale -DUCE_CHAIN_DEBUG object into it).
VARIANT = "" if FLAGS == BASE_FLAGS else hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:10]   # ANY difference (-D, -O1, -g, -mllvm ...)
OBJ_DIR = os.path.join(LIB_DIR, "obj", VARIANT or "product")
LIB_PATH = os.path.join(LIB_DIR, "libuce_hip.so" if not VARIANT else f"libuce_hip.{VARIANT}.so")


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [PUBLIC_HEADER]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def source_hash() -> str:
    """Short hash of every kernel source and header: which state of csrc/ a measurement belongs to (profiles/traffic.json
    stamps it next to the folded counters; bench.py flags entries collected on another state as stale)."""
    h = hashlib.sha256()
    for path in sources() + headers():
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    """True when the library is missing or older than ANY csrc/*.hip, csrc/*.h or the public header."""
    flags_file = os.path.join(OBJ_DIR, ".flags")
    if not os.path.exists(flags_file) or open(flags_file).read() != " ".join(FLAGS):
        return True                            # the objects this library was linked from were compiled with other flags
    return _stale(LIB_PATH, sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags_file = os.path.join(OBJ_DIR, ".flags")
    flag_line = " ".join(FLAGS)
    if not os.path.exists(flags_file) or open(flags_file).read() != flag_line:     # objects of another flag set: all stale
        force = True
        with open(flags_file, "w") as fh:
            fh.write(flag_line)
    cc, hdrs = _hipcc(), headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def compile_one(src: str) -> None:
        cmd = [cc] + FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print("[uce_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, todo))
    objs = [_obj(s) for s in sources()]
    stale_objs = set(glob.glob(os.path.join(OBJ_DIR, "*.o"))) - set(objs)
    for o in stale_objs:                      # a removed source must not stay linked in
        os.remove(o)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + LINK_FLAGS + ["-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print("[uce_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of the compiler behind hipcc (to LD_PRELOAD into a Python that loads the ASAN build)."""
    out = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        cands = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        if not cands:
            raise RuntimeError("libclang_rt.asan-x86_64.so not found")
        out = cands[-1]
    return out


def asan_lib_path() -> str:
    """Where build_asan() puts its library (the variant name is a hash of the flag set)."""
    flags = BASE_FLAGS + ASAN_DEFINES.split()
    return os.path.join(LIB_DIR, "libuce_hip.%s.so" % hashlib.sha256(" ".join(flags).encode()).hexdigest()[:10])


def build_asan(verbose: bool = False) -> str:
    """Build (if stale) the host-ASAN variant in a child interpreter - the flag set is fixed at import - and return its path."""
    import sys
    env = dict(os.environ, UCE_DEFINES=ASAN_DEFINES)
    env.pop("UCE_CHAIN_DEBUG", None)
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "--if-stale"] + ([] if verbose else ["--quiet"]),
                         env=env, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("ASAN build failed:\n" + res.stdout[-2000:] + res.stderr[-2000:])
    return res.stdout.strip().splitlines()[-1]


if __name__ == "__main__":
    import sys
    build(force="--if-stale" not in sys.argv, verbose="--quiet" not in sys.argv)
    print(LIB_PATH)
