"""Host-side timeline of ONE decoded batch (240 photo-like 1600x900 JPEGs from the pinned arena): when the submitting
thread parses headers, enters / leaves every group's _submit and _finish.  Usage: python tools/jpeg_host_timeline.py"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from cama_amd import jpeg as J

rng = np.random.default_rng(0)
y, x = np.mgrid[0:900, 0:1600]
base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
blobs = []
for _ in range(240):
    b = io.BytesIO(); Image.fromarray(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)).save(b, format="JPEG", quality=90)
    blobs.append(b.getvalue())
dec = J.DeviceJpegDecoder("cuda:0")
staged = dec.stage(blobs)
out = dec.decode(staged)
for _ in range(3):
    dec.decode(staged, out=out)
torch.cuda.synchronize()
log = []
for name in ("_submit", "_finish"):
    f = getattr(dec, name)
    def wrap(*a, _f=f, _n=name, **k):
        t = time.perf_counter(); r = _f(*a, **k); log.append((_n, t, time.perf_counter())); return r
    setattr(dec, name, wrap)
ph = J.parse_header_blob
tp = [0.0]
def parse(b):
    t = time.perf_counter(); r = ph(b); tp[0] += time.perf_counter() - t; return r
J.parse_header_blob = parse
for rep in range(2):
    log.clear(); tp[0] = 0.0
    t0 = time.perf_counter()
    pend = dec.decode_async(staged, out=out)
    t1 = time.perf_counter()
    pend.result()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: decode_async returns at {(t1 - t0) * 1e6:.0f} us (header parsing {tp[0] * 1e6:.0f} us), result at {(t2 - t0) * 1e6:.0f} us")
    print("   " + "  ".join(f"{n[1:]}[{(a - t0) * 1e6:.0f}-{(b - t0) * 1e6:.0f}]" for n, a, b in log))
