"""ctypes binding of ``libmirec.so`` (declarations: ``include/mirec.h``).

There is no CPU fallback: ``require()`` raises if the shared object is missing,
and every wrapper raises ``RuntimeError`` with the library's own message when a
call returns an error code, so a model plugin's ``except Exception`` fallback
(reference ``recommenders/als_recommender.py:203-207``) keeps working.
"""

from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmirec.so")

_c = ctypes
_vp = _c.c_void_p
_i64 = _c.c_int64
_i32 = _c.c_int32
_int = _c.c_int
_f32 = _c.c_float
_sz = _c.c_size_t

# name -> (restype, argtypes); mirrors include/mirec.h one to one
SIGNATURES = {
    "mirec_version": (_int, []),
    "mirec_last_error": (_c.c_char_p, []),
    "mirec_factor_width": (_int, [_int, _int]),
    "mirec_gram_workspace_bytes": (_sz, [_i64, _int]),
    "mirec_gram": (_int, [_vp, _i64, _int, _f32, _vp, _vp, _vp, _sz, _vp]),
    "mirec_gram_static_max_bytes": (_sz, []),
    "mirec_gram_static": (_int, [_vp, _i64, _int, _f32, _vp, _vp, _vp]),
    "mirec_als_plan_bytes": (_sz, [_i64, _i64, _i32]),
    "mirec_als_short_cache_rows": (_int, [_int]),
    "mirec_als_plan_build": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _sz, _c.POINTER(_i32), _i32,
                                    _c.POINTER(_i64), _vp]),
    "mirec_als_sweep_workspace_bytes": (_sz, [_c.POINTER(_i64), _int]),
    "mirec_als_half_sweep": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _int, _int, _vp, _i32, _i32,
                                    _c.POINTER(_i64), _f32, _vp, _sz, _vp]),
    "mirec_csr_transpose_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mirec_csr_transpose": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mirec_lightfm_warp_workspace_bytes": (_sz, [_int]),
    "mirec_lightfm_warp_epoch": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_double, _c.c_double,
                                        _c.c_double, _int, _c.c_uint32, _int, _i64, _vp, _sz, _vp]),
    "mirec_lightfm_ranks": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mirec_topk_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "mirec_topk": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "mirec_topk_order_workspace_bytes": (_sz, [_i64]),
    "mirec_topk_order_build": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mirec_topk_ordered": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "mirec_topk_small_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "mirec_topk_small": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "mirec_lightfm_repr": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _f32, _int, _vp, _int, _int, _int, _int, _vp]),
    "mirec_lightfm_predict": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "mirec_events_to_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "mirec_events_to_csr": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mirec_ranking_hits": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
}

ABI_VERSION = 102   # mirec_version() of the library these signatures belong to

_lib = None
_lock = threading.Lock()


class MirecError(RuntimeError):
    """A libmirec call returned an error code."""


def load(path: str = LIB_PATH):
    """dlopen the library and attach the prototypes (no GPU needed)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(path):
                raise MirecError(
                    f"{path} is missing: build it with `python __graft_entry__.py` "
                    "(hipcc --offload-arch=gfx950); there is no CPU fallback for the recommender core")
            lib = ctypes.CDLL(path)
            lib.mirec_version.restype = _int
            lib.mirec_version.argtypes = []
            have = lib.mirec_version()
            if have < ABI_VERSION:
                # entry points have changed their parameters: an older build would be called with shifted arguments
                raise MirecError(f"{path} is version {have}, this package binds version {ABI_VERSION}: rebuild it "
                                 "with `python __graft_entry__.py`")
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def require():
    """The loaded library, or a loud failure."""
    return load()


def last_error() -> str:
    msg = load().mirec_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise MirecError(f"{what or 'libmirec call'} failed (code {rc}): {last_error()}")
