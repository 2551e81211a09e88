"""Device JPEG decoder throughput: batches of 1600x900 frames, worst-case noise and photo-like content.
Usage: python tools/jpeg_probe.py [--batch 6] [--reps 20]"""
import argparse, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from cama_amd.jpeg import DeviceJpegDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--restart-rows", type=int, default=0, help="encode with a restart interval of this many MCU rows")
ap.add_argument("--lanes", type=int, default=0, help="fixed number of groups (0 = the decoder's own sizing)")
ap.add_argument("--min-group", type=int, default=8)
ap.add_argument("--sets", default="noise,photo", help="which image sets to run")
a = ap.parse_args()
rng = np.random.default_rng(0)
y, x = np.mgrid[0:900, 0:1600]
base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)


def enc(im, q=90):
    kw = dict(restart_marker_rows=a.restart_rows) if a.restart_rows else {}
    b = io.BytesIO(); Image.fromarray(im).save(b, format="JPEG", quality=q, **kw); return b.getvalue()


make = {"noise": lambda: enc(rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8)),
        "photo": lambda: enc(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8))}
sets = {k: [make[k]() for _ in range(a.batch)] for k in make if k in a.sets.split(",")}
dec = DeviceJpegDecoder("cuda:0", lanes=a.lanes or None, min_group=a.min_group)
for name, blobs in sets.items():
    out = dec.decode(blobs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.reps):
        out = dec.decode(blobs, out=out)
    torch.cuda.synchronize()
    dt_mem = (time.perf_counter() - t) / a.reps
    # the pipeline's path: the files already sit in a pinned arena (ClipFrameSource's readers put them there), the
    # decoder uploads the span they occupy -- no packing copy on the submitting thread
    staged = dec.stage(blobs)
    assert torch.equal(dec.decode(staged), out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.reps):
        out = dec.decode(staged, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.reps
    print(f"{name}: from bytes objects (packed into a staging buffer per call): {dt_mem * 1e3:.2f} ms/batch = "
          f"{a.batch / dt_mem:.0f} images/s; from the pinned arena:")
    t = time.perf_counter()
    for b in blobs[:2]:
        np.array(Image.open(io.BytesIO(b)).convert("RGB"))
    host = (time.perf_counter() - t) / 2
    print(f"{name}: {sum(map(len, blobs)) / len(blobs) / 1e3:.0f} KB/image, batch {a.batch}: {dt * 1e3:.2f} ms/batch = "
          f"{a.batch / dt:.0f} images/s = {a.batch / dt / 6:.0f} six-camera frames/s   (Pillow, 1 core: {host * 1e3:.1f} ms/image)")
print(dec.stats)
