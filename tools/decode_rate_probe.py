import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from cama_amd.frames import read_rgb_or_bgr
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
for i in range(48):
    p = os.path.join(d, f"{i}.jpg"); Image.fromarray(rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8)).save(p, quality=90); paths.append(p)
# smoother, more realistic content decodes faster than noise; also try a gradient image
g = np.linspace(0, 255, 1600)[None, :, None] * np.ones((900, 1, 3)); gp = os.path.join(d, "g.jpg"); Image.fromarray(g.astype(np.uint8)).save(gp, quality=90)
t = time.perf_counter(); [read_rgb_or_bgr(p) for p in paths[:12]]; print(f"1 thread, noise JPEG (1.3 MB): {(time.perf_counter()-t)/12*1e3:.1f} ms/image")
t = time.perf_counter(); [read_rgb_or_bgr(gp) for _ in range(12)]; print(f"1 thread, smooth JPEG ({os.path.getsize(gp)//1024} KB): {(time.perf_counter()-t)/12*1e3:.1f} ms/image")
for w in (6, 12, 24, 48):
    with ThreadPoolExecutor(w) as ex:
        t = time.perf_counter(); list(ex.map(read_rgb_or_bgr, paths * 2)); dt = time.perf_counter() - t
    print(f"{w} threads: {96/dt:.0f} images/s = {96/dt/6:.0f} six-camera frames/s")
