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
