"""Where k_jpeg_sync<1> spends its time, per workgroup: needs a library built with -DJPEG_TRACE (tools/jpeg_phase_clock.sh).
Prints, over the workgroups of the last launches, the time of the setup, the first (speculative) decode and every fixpoint
round (100 MHz constant clock), and how many subsequences each round re-decoded."""
import ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from cama_amd import _lib
from cama_amd.jpeg import DeviceJpegDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 34
rng = np.random.default_rng(0)
y, x = np.mgrid[0:900, 0:1600]
base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
blobs = []
for _ in range(n):
    b = io.BytesIO(); Image.fromarray(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)).save(b, format="JPEG", quality=90)
    blobs.append(b.getvalue())
dec = DeviceJpegDecoder("cuda:0")
staged = dec.stage(blobs)
out = dec.decode(staged)
for _ in range(3):
    dec.decode(staged, out=out)
torch.cuda.synchronize()
lib = _lib.lib()
clk = np.zeros((4096, 16), np.uint64); red = np.zeros((4096, 16), np.uint32)
f = lib.cama_diag_jpeg_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert f(clk.ctypes.data, red.ctypes.data) == 0
used = clk[:, 0] > 0
c = clk[used].astype(np.int64); m = red[used]
print(f"{n} images, {used.sum()} traced workgroups (the last launch that used each block index)")
us = lambda t: t / 100.0
print(f"setup           median {np.median(us(c[:, 1] - c[:, 0])):7.1f} us   max {us(c[:, 1] - c[:, 0]).max():7.1f}")
print(f"first decode    median {np.median(us(c[:, 2] - c[:, 1])):7.1f} us   max {us(c[:, 2] - c[:, 1]).max():7.1f}")
for r in range(13):
    a, b = c[:, 2 + r], c[:, 3 + r]
    ok = (b > 0) & (a > 0) & (m[:, 2 + r] > 0)
    if not ok.any():
        continue
    d = us(b[ok] - a[ok])
    print(f"round {r:2d}: {ok.sum():4d} workgroups, re-decoded median {np.median(m[ok, 2 + r]):5.0f} max {m[ok, 2 + r].max():3d}; "
          f"median {np.median(d):7.1f} us   max {d.max():7.1f}")
last = np.array([row[row > 0].max() for row in c])
tot = us(last - c[:, 0])
print(f"whole workgroup median {np.median(tot):7.1f} us   max {tot.max():7.1f};  launch span (first entry -> last exit) {us(last.max() - c[:, 0].min()):7.1f} us")
