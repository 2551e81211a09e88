"""cProfile of the host side of DeviceJpegDecoder.decode on a 240-image photo-like batch staged in a pinned arena."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from cama_amd.jpeg import DeviceJpegDecoder
rng = np.random.default_rng(0)
y, x = np.mgrid[0:900, 0:1600]
base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
def enc(im):
    b = io.BytesIO(); Image.fromarray(im).save(b, format="JPEG", quality=90); return b.getvalue()
blobs = [enc(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)) for _ in range(24)] * 10
blobs = blobs[:int(os.environ.get('BATCH', 240))]
dec = DeviceJpegDecoder("cuda:0")
staged = dec.stage(blobs)
out = dec.decode(staged)
for _ in range(3): dec.decode(staged, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): dec.decode(staged, out=out)
torch.cuda.synchronize()
print(f"decode: {(time.perf_counter() - t) / 10 * 1e3:.2f} ms per {len(blobs)}")
t = time.perf_counter()
for _ in range(10): p = dec.decode_async(staged, out=out)
t1 = time.perf_counter(); p.result(); torch.cuda.synchronize()
print(f"decode_async issue only: {(t1 - t) / 10 * 1e3:.2f} ms per {len(blobs)}")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): p = dec.decode_async(staged, out=out)
pr.disable(); p.result()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
