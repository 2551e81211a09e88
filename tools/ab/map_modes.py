import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
a = bench.parse_args([])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
job = bench.Job(a, [0], dev, None)
by = 36 * a.width * a.height * a.frames
res = []
for rnd in range(2):
    for m in ("31", "0", "5", "8", "10"):
        os.environ["CAMA_OVERLAY_CHUNK_LOG2"] = m
        for _ in range(4): job.step()
        job.eng.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): job.step()
        job.eng.join(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        res.append("%s:%.3f" % (m, by / dt / 8e12))
print(" ".join(res))
