import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from cama_amd.frames import read_rgb_or_bgr
d = os.environ.get("CAMA_PROBE_DIR") or tempfile.mkdtemp()
os.environ["CAMA_PROBE_DIR"] = d
paths = [os.path.join(d, f"{i}.jpg") for i in range(48)]


def _main_threads():
    rng = np.random.default_rng(0)
    for p in paths:
        Image.fromarray(rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8)).save(p, quality=90)
    # smoother, more realistic content decodes faster than noise; also try a gradient image
    g = np.linspace(0, 255, 1600)[None, :, None] * np.ones((900, 1, 3)); gp = os.path.join(d, "g.jpg"); Image.fromarray(g.astype(np.uint8)).save(gp, quality=90)
    t = time.perf_counter(); [read_rgb_or_bgr(p) for p in paths[:12]]; print(f"1 thread, noise JPEG (1.3 MB): {(time.perf_counter()-t)/12*1e3:.1f} ms/image")
    t = time.perf_counter(); [read_rgb_or_bgr(gp) for _ in range(12)]; print(f"1 thread, smooth JPEG ({os.path.getsize(gp)//1024} KB): {(time.perf_counter()-t)/12*1e3:.1f} ms/image")
    for w in (6, 12, 24, 48):
        with ThreadPoolExecutor(w) as ex:
            t = time.perf_counter(); list(ex.map(read_rgb_or_bgr, paths * 2)); dt = time.perf_counter() - t
        print(f"{w} threads: {96/dt:.0f} images/s = {96/dt/6:.0f} six-camera frames/s")


# process pool decoding straight into a shared-memory slab (what ClipFrameSource(workers="process") does)
import multiprocessing as mp
from multiprocessing import shared_memory


def _decode_into(args):
    name, slot, path = args
    shm = shared_memory.SharedMemory(name=name)
    try:
        dst = np.ndarray((900, 1600, 3), np.uint8, buffer=shm.buf, offset=slot * 900 * 1600 * 3)
        with Image.open(path) as im:
            im.draft("RGB", (1600, 900))
            dst[...] = np.asarray(im.convert("RGB"))
    finally:
        shm.close()
    return slot


if __name__ == "__main__":
    _main_threads()
    n_img = 192
    shm = shared_memory.SharedMemory(create=True, size=n_img * 900 * 1600 * 3)
    try:
        for procs in (8, 16, 32, 64, 128):
            if procs > (os.cpu_count() or 1):
                break
            with mp.get_context("spawn").Pool(procs) as pool:
                pool.map(_decode_into, [(shm.name, i, paths[i % 48]) for i in range(procs)])          # warm the workers
                t = time.perf_counter()
                pool.map(_decode_into, [(shm.name, i, paths[i % 48]) for i in range(n_img)], chunksize=1)
                dt = time.perf_counter() - t
            print(f"{procs} processes -> shared memory: {n_img/dt:.0f} images/s = {n_img/dt/6:.0f} six-camera frames/s")
    finally:
        shm.close()
        shm.unlink()
