"""ctypes binding of ``libgd_raster.so`` (C-ABI declared in ``include/gd_raster.h``).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every
entry point raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgd_raster.so")
_lib = None


def use_library(path: str) -> None:
    """Point the binding at another build of libgd_raster.so BEFORE its first use -- same-box A/B timing of experimental
    builds by the scripts under tools/ (tools/ablib.py) and ``bench.py --raster-lib``; the package itself reads no
    environment variable for this."""
    global _LIB_PATH
    if _lib is not None:
        raise RuntimeError("libgd_raster.so is already loaded")
    _LIB_PATH = os.path.abspath(path)


GD_MAX_VIEWS = 16
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float


class Layout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "point_offsets",
        "block_sums", "ranges", "n_contrib", "pair_counts", "point_list", "point_list_alt", "keys", "keys_alt", "sort_hist")]


# symbol -> (restype, argtypes); mirrors include/gd_raster.h one to one
_PF = C.POINTER(_f)
SIGNATURES = {
    "gd_raster_geom_bytes": (C.c_size_t, [_i, _i]),
    "gd_raster_image_bytes": (C.c_size_t, [_i, _i, _i]),
    "gd_raster_binning_bytes": (C.c_size_t, [C.c_int64]),
    "gd_raster_backward_scratch_bytes": (C.c_size_t, [_i, _i, C.c_int64]),
    "gd_raster_forward": (_i, [_vp] + [ALLOC_FN, _vp] * 3      # stream, 3 x (allocator, user)
                          + [_i] * 3 + [_vp] + [_i] * 2          # P D M, background, width height
                          + [_vp] * 5 + [_f] + [_vp] * 5         # means3D shs colors opac scales | mod | rot cov view proj campos
                          + [_f] * 2 + [_i] + [_vp] * 4 + [_i]), # tanx tany, prefiltered, color depth alpha radii, debug
    "gd_raster_backward": (_i, [_vp] + [_i] * 4 + [_vp] + [_i] * 2  # stream, P D M R, background, width height
                           + [_vp] * 5 + [_f] + [_vp] * 5            # means3D shs colors alphas scales | mod | rot cov view proj campos
                           + [_f] * 2 + [_vp] * 5                     # tanx tany, radii geom binning image bwd_scratch
                           + [_vp] * 3 + [_vp] * 10 + [_i]),          # dL_dpix/depth/alpha, 10 outputs, debug
    "gd_raster_mark_visible": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "gd_raster_forward_batched": (_i, [_vp, _i] + [ALLOC_FN, _vp] * 3
                                  + [_i] * 3 + [_vp] + [_i] * 2
                                  + [_vp] * 5 + [_f] + [_vp] * 5
                                  + [_PF] * 2 + [_i] + [_vp] * 4 + [_i]),
    "gd_raster_forward_batched_capacity": (_i, [_vp, _i] + [ALLOC_FN, _vp] * 3
                                           + [_i] * 3 + [_vp] + [_i] * 2
                                           + [_vp] * 5 + [_f] + [_vp] * 5
                                           + [_PF] * 2 + [_i] + [_vp] * 4 + [_i] + [C.c_int64, _vp]),   # + capacity, count_dev
    "gd_raster_backward_batched": (_i, [_vp, _i] + [_i] * 4 + [_vp] + [_i] * 2
                                   + [_vp] * 5 + [_f] + [_vp] * 5
                                   + [_PF] * 2 + [_vp] * 5
                                   + [_vp] * 3 + [_vp] * 8 + [_i]),   # 8 outputs (no dL_dconic / dL_ddepth)
    "gd_raster_get_layout": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, C.POINTER(Layout)]),
    "gd_raster_sort_bits": (_i, [_i, _i, _i]),
    "gd_raster_blend_exp": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i]),
    "gd_raster_profile_enable": (_i, [_i]),
    "gd_raster_profile_collect": (_i, []),
    "gd_raster_force_binning": (_i, [_i]),
    "gd_raster_poison_lds": (_i, [_vp]),
    "gd_raster_profile_get": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "gd_raster_profile_reset": (_i, []),
    "gd_raster_profile_kernel_name": (C.c_char_p, [_i]),
    "gd_raster_last_error": (C.c_char_p, []),
    "gd_raster_build_info": (C.c_char_p, []),
}
# scene-side kernels in the same library (include/gd_scene.h)
_I64P = C.POINTER(C.c_int64)
SCENE_SIGNATURES = {
    "gd_scene_dist2_scratch_bytes": (C.c_size_t, [_i]),
    "gd_scene_dist2": (_i, [_vp, _i, _vp, _vp, _vp]),
    "gd_scene_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _I64P, C.POINTER(C.c_double), C.c_double,
                                C.c_double, C.c_double, _i]),
    "gd_scene_densify_stats": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "gd_scene_activate_forward": (_i, [_vp, _i, _i] + [_vp] * 9),
    "gd_scene_activate_backward": (_i, [_vp, _i, _i] + [_vp] * 12),
    "gd_scene_densify_scratch_bytes": (C.c_size_t, [_i]),
    "gd_scene_densify_plan": (_i, [_vp, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp,
                                   C.POINTER(C.c_uint32)]),
    "gd_scene_densify_apply": (_i, [_vp, _i, _i, C.POINTER(C.c_int), _i, _i, _i, C.POINTER(C.c_uint32)] + [_vp] * 8),
    "gd_scene_densify_last_error": (C.c_char_p, []),
    "gd_scene_last_error": (C.c_char_p, []),
}


class NativeLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def lib():
    """Load libgd_raster.so (once).  Raises NativeLibraryError if it is absent -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_PATH} not found: build it with `python -m garmentdreamer_amd._build` "
                "(hipcc --offload-arch=gfx950). The HIP rasterizer has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; load it FIRST so
        # our library's DT_NEEDED libamdhip64.so.7 binds to the same instance (streams and device
        # pointers are shared with torch).  Loading ours first puts a second runtime in the
        # process and every call then fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        try:
            L = C.CDLL(_LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise NativeLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        for name, (res, args) in list(SIGNATURES.items()) + list(SCENE_SIGNATURES.items()):
            fn = getattr(L, name)  # AttributeError here == ABI drift; let it surface
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ret: int, what: str) -> int:
    if ret < 0:
        msg = lib().gd_raster_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({ret}): {msg}")
    return ret


def check_scene(ret: int, what: str, err: str = "gd_scene_last_error") -> int:
    if ret < 0:
        raise RuntimeError(f"{what} failed ({ret}): {getattr(lib(), err)().decode('utf-8', 'replace')}")
    return ret


KERNEL_IDS = {"preprocess": 0, "scan": 1, "duplicate": 2, "sort": 3, "ranges": 4, "render_fwd": 5, "render_bwd": 6,
              "preprocess_bwd": 7}


def profile_enable(on: bool) -> None:
    lib().gd_raster_profile_enable(int(on))


def profile_reset() -> None:
    lib().gd_raster_profile_reset()


def profile_read() -> dict:
    """{kernel: (total_ms, launches)} after waiting for all recorded events."""
    L = lib()
    L.gd_raster_profile_collect()
    out = {}
    for name, kid in KERNEL_IDS.items():
        ms, n = C.c_double(0), C.c_int64(0)
        L.gd_raster_profile_get(kid, C.byref(ms), C.byref(n))
        out[name] = (ms.value, n.value)
    return out
