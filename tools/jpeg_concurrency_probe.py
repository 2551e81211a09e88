"""Is the chip saturated by one decoded batch?  Aggregate rate of K batches of 240 photo-like 1600x900 JPEGs in flight at once
(decode_async x K, then result() x K) against one batch at a time, and the one-batch rate for several group counts.
Usage: GPU_MAX_HW_QUEUES=16 python tools/jpeg_concurrency_probe.py"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from cama_amd.jpeg import DeviceJpegDecoder

rng = np.random.default_rng(0)
y, x = np.mgrid[0:900, 0:1600]
base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
blobs = []
for _ in range(240):
    b = io.BytesIO(); Image.fromarray(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)).save(b, format="JPEG", quality=90)
    blobs.append(b.getvalue())
dec = DeviceJpegDecoder("cuda:0")
K = 3
staged = [dec.stage(blobs) for _ in range(K)]
outs = [dec.decode(s) for s in staged]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for groups in [int(g) or None for g in os.environ.get("PROBE_GROUPS", "0,4,7,10,14,20").split(",")]:
    for k in (1, 2, 3):
        for _ in range(2):
            [p.result() for p in [dec.decode_async(staged[i], out=outs[i], groups=groups) for i in range(k)]]
        torch.cuda.synchronize()
        reps = 5
        t = time.perf_counter()
        for _ in range(reps):
            pend = [dec.decode_async(staged[i], out=outs[i], groups=groups) for i in range(k)]
            [p.result() for p in pend]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(f"groups per batch {groups}: {k} batch(es) in flight: {dt * 1e3:.2f} ms = {240 * k / dt:.0f} images/s")
