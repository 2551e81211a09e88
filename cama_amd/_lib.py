"""ctypes binding of libcama_hip.so (include/cama_hip.h).

The HIP library IS the product: there is no CPU fallback.  Importing this module
never fails (so host-only code paths and the CPU test-suite can import the
package), but the first call to `lib()` raises if the shared object is missing.
"""
import ctypes
import os
from os.path import dirname, join, abspath, exists

_HERE = dirname(abspath(__file__))
LIB_PATH = join(_HERE, "libcama_hip.so")
if os.environ.get("CAMA_HIP_LIB") and os.environ.get("CAMA_ALLOW_LIB_OVERRIDE") == "1":
    LIB_PATH = os.environ["CAMA_HIP_LIB"]          # another build of the library: kernel A/B experiments (tools/) only


def test_hooks():
    """CAMA_TEST_HOOKS = "name[=value],..." as a dict (the parity suite's child processes force production code paths with it:
    the library reads the same variable, cama_hip.hip: test_hook).  Python-side names: bounds_min_verts=n, no_bounds."""
    out = {}
    for item in os.environ.get("CAMA_TEST_HOOKS", "").split(","):
        if item:
            k, _, v = item.partition("=")
            out[k] = int(v) if v else 1
    return out
ABI_VERSION = 24
BIN_WORKLIST = 1
BIN_SEGMENTS = 2
BIN_SEGMENTS_WU = 4            # EXTENSION: discs + one-pixel segments between polyline neighbours (include/cama_hip.h)

_vp, _i32, _i64, _sz, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
_u64 = ctypes.c_uint64

MAX_RADIUS = 15


class Clip(ctypes.Structure):
    """include/cama_hip.h: cama_clip -- what stays the same from launch to launch of one clip."""
    _fields_ = [("x", _vp), ("y", _vp), ("z", _vp), ("colour_id", _vp), ("draw_key", _vp), ("block_bounds", _vp),
                ("c2cam", _vp), ("K", _vp), ("vrows", _vp), ("band_rows", _vp), ("N", _i64), ("crop", ctypes.c_double * 6),
                ("xyz_is_f64", _i32), ("flags", _i32), ("C", _i32), ("W", _i32), ("H", _i32), ("cols", _i32), ("radius", _i32),
                ("kind", _i32), ("H0", _i32), ("W0", _i32), ("max_src_rows", _i32), ("reserved", _i32),
                ("halfwidth", _i32 * (MAX_RADIUS + 1)), ("palette_bgr", ctypes.c_uint8 * 8)]


# name -> (restype, argtypes); mirrors include/cama_hip.h one to one
SIGNATURES = {
    "cama_abi_version": (_i32, []),
    "cama_last_error": (ctypes.c_char_p, []),
    "cama_transform_points": (_i32, [_vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "cama_crop_points": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "cama_project_points": (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cama_project_frames": (_i32, [_vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32,
                                   _vp, _vp, _vp, _vp]),
    "cama_render_scratch_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32, _i32]),
    "cama_map_bounds_block": (_i32, []),
    "cama_map_bounds": (_i32, [_vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "cama_render_frames": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32,
                                  _vp, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_bin_frames": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32,
                               _vp, _sz, _vp]),
    "cama_overlay_frames": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_build_static_map": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _i32, _i32,
                                     _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "cama_resample_frames": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "cama_overlay_frames_alpha": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "cama_overlay_frames_raw": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _vp, _vp, _vp, _sz, _vp]),
    "cama_bgr_to_i420": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "cama_raw35_plan": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cama_overlay_frames_raw35": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                         _vp, _vp, _vp, _sz, _vp]),
    "cama_pipeline_create": (_i32, [_vp]),
    "cama_pipeline_destroy": (_i32, [_vp]),
    "cama_pipeline_render": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32,
                                    _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cama_pipeline_render_raw35": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i32,
                                          _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cama_bin_scenes": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "cama_overlay_scenes": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_render_scenes": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_pipeline_render_scenes": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp,
                                           _vp, _vp, _sz, _vp]),
    "cama_pipeline_stage_poses": (_i32, [_vp, _vp, _i32, _vp]),
    "cama_pipeline_render_clip": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "cama_pipeline_join": (_i32, [_vp, _vp]),
    "cama_pipeline_issued": (_i64, [_vp]),
    "cama_pipeline_completed": (_i64, [_vp]),
    "cama_pipeline_scratch_bytes": (_i64, [_vp]),
    "cama_pipeline_info": (_i32, [_vp, _vp]),
    "cama_pipeline_bin_stats": (_i32, [_vp, _vp]),
    "cama_pipeline_guard_check": (_i32, [_vp, _vp]),
    "cama_stamp_scratch_bytes": (_sz, [_i32, _i32]),
    "cama_stamp_points": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_probe_xcd_map": (_i32, [_vp, _i32, _vp]),
    "cama_overlay_mapping_info": (_i32, [ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.POINTER(ctypes.c_double)]),
    "cama_overlay_probe": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_double), _vp]),
    "cama_set_option": (_i32, [ctypes.c_char_p, _i64]),
    "cama_get_option": (_i32, [ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "cama_stamp_polylines": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_stamp_polylines_wu": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cama_circle_halfwidths": (_i32, [_i32, _vp]),
    "cama_overlay_band_rows": (_i32, [_i32]),
    "cama_jpeg_image_bytes": (_sz, []),
    "cama_jpeg_huff_set_bytes": (_sz, []),
    "cama_jpeg_plan": (_i32, [_vp, _i32, _u64, _vp]),
    "cama_jpeg_find_restarts": (_i32, [_vp, _u64, _vp, ctypes.c_uint32, _vp, _vp]),
    "cama_jpeg_decode": (_i32, [_vp, _u64, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _u64, _i32, _vp, _sz, _vp, _vp]),
    "cama_read_files": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "cama_profile_enable": (_i32, [_i32]),
    "cama_profile_collect": (_i32, [_vp, _vp]),
    "cama_profile_collect_each": (_i32, [_vp, _i32, _vp]),
    "cama_profile_collect_project": (_i32, [_vp, _vp]),
    "cama_bin_stats": (_i32, [_vp, _sz, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
}

# the entries include/cama_hip_diag.h declares (diagnostics, live timing, options, test hooks): not part of the contract
DIAG = ("cama_pipeline_info", "cama_pipeline_bin_stats", "cama_pipeline_guard_check", "cama_probe_xcd_map",
        "cama_overlay_mapping_info", "cama_set_option", "cama_get_option", "cama_profile_enable", "cama_profile_collect",
        "cama_profile_collect_each", "cama_profile_collect_project", "cama_bin_stats")

_lib = None


class CamaHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not exists(LIB_PATH):
            raise CamaHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  cama_amd has no CPU fallback.")
        # torch ships its own libamdhip64.so.7; load it first so this library binds to the SAME HIP runtime
        # (two runtimes in one process do not share devices, streams or allocations)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.cama_abi_version() != ABI_VERSION:
            raise CamaHipError(f"ABI mismatch: library {L.cama_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


ENOMEM = -3                    # CAMA_ENOMEM: a pipeline could not grow its own scratch


def check(rc):
    if rc != 0:
        msg = f"libcama_hip error {rc}: {lib().cama_last_error().decode()}"
        if rc == ENOMEM:
            # the same exception torch's allocator raises, so the callers' out-of-memory handling (ClipManager.render_clip:
            # halve the launch) covers memory the library allocates itself
            import torch
            raise getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError)(msg)
        raise CamaHipError(msg)


def call_retrying_oom(fn, *args):
    """fn(*args) -> check(); on CAMA_ENOMEM release what torch's caching allocator has parked (hipMalloc inside the library
    cannot use it) and try once more before raising torch.OutOfMemoryError."""
    rc = fn(*args)
    if rc == ENOMEM:
        import torch
        torch.cuda.empty_cache()
        rc = fn(*args)
    check(rc)


def circle_halfwidths(radius):
    import numpy as np
    hw = np.zeros(radius + 1, np.int32)
    n = lib().cama_circle_halfwidths(radius, hw.ctypes.data)
    if n < 0:
        check(n)
    return hw
