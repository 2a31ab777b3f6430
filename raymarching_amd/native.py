"""ctypes binding of librmhip.so -- the one shared library of this package.

The library is built in-tree (``raymarching_amd/librmhip.so``) by
``__graft_entry__.build()`` / ``make -C raymarching_amd/csrc``.  There is no
fallback: a missing library raises at import of this module's `lib()`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMHIP_LIB") or os.path.join(_HERE, "librmhip.so")   # RMHIP_LIB: developer override for A/B builds

RM_OK, RM_E_ARG, RM_E_DEVICE, RM_E_NOMEM, RM_E_RANGE = 0, -1, -2, -3, -4
RM_SPHERE, RM_BOX, RM_LINK, RM_CYLINDER, RM_TORUS = range(5)
RM_MAX_PRIMS = 1024


class RmPrimitive(C.Structure):
    """rm_primitive of include/rmhip.h"""
    _fields_ = [("type", C.c_int32), ("op", C.c_int32), ("pos", C.c_float * 3), ("quat", C.c_float * 4),
                ("scale", C.c_float * 3), ("color", C.c_float * 3), ("params", C.c_float * 4)]


class RmStats(C.Structure):
    """rm_stats of include/rmhip.h"""
    _fields_ = [("n_pixels", C.c_uint64), ("n_hit", C.c_uint64), ("sum_steps_p", C.c_uint64),
                ("sum_steps_s", C.c_uint64)]


class RmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"librmhip error {code}: {msg}")
        self.code = code


# every symbol include/rmhip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_FP = C.POINTER(C.c_float)
SYMBOLS = [
    ("rm_create", _P, [C.c_int]),
    ("rm_destroy", None, [_P]),
    ("rm_last_error", C.c_char_p, [_P]),
    ("rm_version", C.c_char_p, []),
    ("UnityPluginLoad", None, [_P]),
    ("UnityPluginUnload", None, []),
    ("rm_set_float", C.c_int, [_P, C.c_char_p, C.c_float]),
    ("rm_set_int", C.c_int, [_P, C.c_char_p, C.c_int]),
    ("rm_set_vector", C.c_int, [_P, C.c_char_p, _FP]),
    ("rm_set_matrix", C.c_int, [_P, C.c_char_p, _FP]),
    ("rm_get_float", C.c_int, [_P, C.c_char_p, _FP]),
    ("rm_get_int", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    ("rm_get_vector", C.c_int, [_P, C.c_char_p, _FP]),
    ("rm_get_matrix", C.c_int, [_P, C.c_char_p, _FP]),
    ("rm_has_property", C.c_int, [_P, C.c_char_p]),
    ("rm_set_texture_device", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int]),
    ("rm_set_primitives", C.c_int, [_P, C.POINTER(RmPrimitive), C.c_int]),
    ("rm_host_alloc", _P, [_P, C.c_size_t]),
    ("rm_host_free", None, [_P, _P]),
    ("rm_blit", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int]),
    ("rm_blit_async", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    ("rm_blit_wait", C.c_int, [_P, C.c_longlong]),
    ("rm_blit_device", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_blit_steps", C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int]),
    ("rm_blit_steps_device", C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_blit_device_noalloc", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_band_rows", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("rm_blit_bands_device", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("rm_blit_poses_device", C.c_int, [_P, C.c_int, _FP, _FP, _FP, _P, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_shadowmap_render", C.c_int, [_P, _P, C.c_int, C.c_int]),
    ("rm_shadowmap_render_device", C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_shadowmap_directional", C.c_int, [_P, _FP, _FP, _P, C.c_int, C.c_int]),
    ("rm_shadowmap_directional_device", C.c_int, [_P, _FP, _FP, _P, _P, C.c_int, C.c_int, _P]),
    ("rm_shadowmap_directional_bands_device", C.c_int, [_P, _FP, _FP, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("rm_pack_rgb_device", C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    ("rm_unpack_rgb_device", C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    ("rm_tile_record_bytes", C.c_size_t, []),
    ("rm_compact_frames_device", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ("rm_expand_frames_device", C.c_int, [_P, _P, C.c_uint32, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("rm_update_frames_device", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("rm_get_stats", C.c_int, [_P, C.POINTER(RmStats)]),
    ("rm_get_leaf_stats", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("rm_debug_counter_slots", C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("rm_set_kernel", C.c_int, [_P, C.c_int]),
]

_lib = None


def lib() -> C.CDLL:
    """Loads librmhip.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C raymarching_amd/csrc` (there is no non-HIP fallback)")
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(ctx, rc: int) -> None:
    if rc != RM_OK:
        msg = lib().rm_last_error(ctx)
        raise RmError(rc, msg.decode() if msg else "")
