"""ctypes binding of libuce_hip.so (include/uce_hip.h).  There is NO CPU fallback: if the
library is missing the product path raises, it never routes through oracle/."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

OK, EINVAL, ENOMEM, EDOM, ENOSYS, ECOMM, ETIMEDOUT = 0, -22, -12, -33, -38, -70, -110
ALGO_AUTO, ALGO_PRIMAL, ALGO_DUAL = 0, 1, 2
DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2
EPILOGUE_NONE, EPILOGUE_GEGLU, EPILOGUE_F32 = 0, 1, 2

_vp, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# symbol -> (restype, argtypes): every entry point include/uce_hip.h declares
SIGNATURES = {
    "uce_version": (_i, []),
    "uce_strerror": (C.c_char_p, [_i]),
    "uce_create": (_i, [C.POINTER(_vp), _i]),
    "uce_destroy": (_i, [_vp]),
    "uce_reserve": (_i, [_vp, _i, _i]),
    "uce_gram": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "uce_solve_delta": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "uce_solve_rhs": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "uce_solve_general": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "uce_apply": (_i, [_vp, _vp, _vp, _vp, _l, _i, _vp]),
    "uce_dual_factors": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "uce_apply_lowrank": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "uce_lowrank_project": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "uce_lowrank_update": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "uce_reserve_rows": (_i, [_vp, _l, _i]),
    "uce_delta_from_factors": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "uce_edit": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _l, _i, _vp]),
    "uce_status": (_i, [_vp, C.POINTER(_i), _vp]),
    "uce_bcast": (_i, [_vp, _vp, _sz, _i, _vp, _vp]),
    "uce_profile_begin": (_i, [_vp]),
    "uce_profile_end": (_i, [_vp, _vp, C.c_char_p, _sz]),
    "uce_debias_targets": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "uce_gather_last_token": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "uce_cast_bf16": (_i, [_vp, _vp, _vp, _l, _vp]),
    "uce_xattn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "uce_sattn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "uce_sattn_packed_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "uce_sattn_exp2_form": (_i, [_vp, _i, _i, _i, _i]),
    "uce_sattn_packed_exp2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "uce_groupnorm_chunks": (_i, [_i]),
    "uce_groupnorm_nhwc_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _l, _vp]),
    "uce_groupnorm_cat_nhwc_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _l, _vp]),
    "uce_add_bias_nhwc_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "uce_cfg_pndm_step": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, C.POINTER(_f), _vp, _f, _f, _vp, _vp, _l, _i, _vp]),
    "uce_geglu_fwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "uce_im2col3x3_nhwc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "uce_im2col3x3_c4": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "uce_conv3x3_nhwc_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "uce_linear_fwd": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _l, _vp, _l, _l, _i, _i, _i, _i, _vp]),
    "uce_linear_colscale_fwd": (_i, [_vp, _vp, _l, _vp, _vp, _l, _l, _i, _i, _i, _f, _i, _vp]),
    "uce_linear_cat_fwd": (_i, [_vp, _vp, _l, _vp, _l, _i, _vp, _vp, _vp, _l, _vp, _l, _l, _i, _i, _i, _i, _vp]),
    "uce_softmax_rows": (_i, [_vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    "uce_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
}

_LIB: Optional[C.CDLL] = None


class UceError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = "?"
        try:
            msg = load().uce_strerror(code).decode()
        except Exception:  # pragma: no cover
            pass
        super().__init__(f"{where} failed: {msg} (code {code})")


def lib_path() -> str:
    return os.environ.get("UCE_HIP_LIB", _build.LIB_PATH)


def load() -> C.CDLL:
    """Load the library (once) and declare the signatures.  Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: the HIP kernels are the product and there is no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `python -m uce_amd.build`.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(code: int, where: str) -> None:
    if code != OK:
        raise UceError(code, where)
