"""CPU: the C-ABI library loads and exports every symbol include/uce_hip.h declares; the
product path fails loudly (no fallback) when there is no GPU."""
import os
import re

import pytest
import torch

from uce_amd import REPO_ROOT
from uce_amd import lib as L


def _declared_functions():
    hdr = open(os.path.join(REPO_ROOT, "include", "uce_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(uce_[a-z0-9_]+)\s*\(", hdr)))


def test_library_is_built_and_loads():
    lib = L.load()
    assert lib.uce_version() >= 100
    assert b"positive definite" in lib.uce_strerror(L.EDOM)


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/uce_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(L.SIGNATURES) == names


def test_argument_validation_without_gpu():
    lib = L.load()
    assert lib.uce_create(None, 0) == L.EINVAL
    assert lib.uce_gram(None, None, None, None, 1, 0, 64, 0.5, None, None, None) == L.EINVAL
    assert lib.uce_apply(None, None, None, None, 1, 64, None) == L.EINVAL
    assert lib.uce_xattn_fwd(None, None, None, None, None, 1, 1, 1, 1, 8, 1.0, 0, None) == L.EINVAL


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a GPU-less box")
def test_no_cpu_fallback():
    from uce_amd import edit
    with pytest.raises(RuntimeError):
        edit.UceHandle("cuda:0")
    with pytest.raises(RuntimeError):
        edit.UceHandle("cpu")


ASAN_CHILD = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["UCE_REPO_ROOT"])
from uce_amd import lib as L
lib = L.load()
assert "asan" not in L.lib_path() and L.lib_path() == os.environ["UCE_HIP_LIB"]
assert lib.uce_version() >= 100
for code in range(-40, 1):                      # every message the table holds, and the out-of-table default
    assert isinstance(lib.uce_strerror(code), bytes)
for hw in (1, 15, 16, 64, 4096, 1 << 20):
    assert 1 <= lib.uce_groupnorm_chunks(hw) <= 64
h = C.c_void_p()
rc = lib.uce_create(C.byref(h), 0)
if rc == 0:                                     # a GPU box: the handle's workspace management under ASAN
    assert lib.uce_reserve_rows(h, 24960, 768) == 0
    assert lib.uce_destroy(h) == 0
else:
    assert rc < 0
# every entry point rejects a null handle before it touches an argument
zero = {C.c_void_p: None, C.c_int: 0, C.c_long: 0, C.c_float: 0.0, C.c_size_t: 0, C.c_char_p: None}
for name, (res, args) in L.SIGNATURES.items():
    if name in ("uce_version", "uce_strerror", "uce_groupnorm_chunks", "uce_create"):
        continue
    if name == "uce_sattn_exp2_form":             # a predicate, not a status: no form without a handle
        assert lib.uce_sattn_exp2_form(None, 0, 0, 0, 0) == 0
        continue
    call = [zero.get(a, None) for a in args]
    rc = getattr(lib, name)(*call)
    assert rc != 0, name
print("asan child ok")
"""


def test_host_side_builds_and_runs_under_address_sanitizer():
    """SURVEY.md section 5 "sanitizers": the host side of the library (handle, workspace, argument checks, dispatch) compiled with
    AddressSanitizer (uce_amd.build.build_asan: -fsanitize=address on the host pass only) loads into a Python started with the
    ASAN runtime preloaded, answers every entry point's argument checks and creates / destroys a handle where there is a GPU -
    with no ASAN report.  The GPU suite runs an edit through the same build (tests/test_stress_gpu.py)."""
    import subprocess
    import sys
    from uce_amd import build as B
    path = B.build_asan()
    assert os.path.exists(path) and path != B.LIB_PATH
    syms = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__asan_init" not in subprocess.run(["nm", "-D", B.LIB_PATH], capture_output=True, text=True).stdout
    env = dict(os.environ, UCE_HIP_LIB=path, LD_PRELOAD=B.asan_runtime(), UCE_REPO_ROOT=REPO_ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86:protect_shadow_gap=0")
    res = subprocess.run([sys.executable, "-c", ASAN_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
    assert res.returncode == 0 and "asan child ok" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-3000:])
