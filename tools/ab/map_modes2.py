import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
argv = sys.argv[1:]
a = bench.parse_args(argv)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
job = bench.Job(a, list(range(max(1, a.scenes))), dev, None)
fh, fw = (900, 1600) if a.raw_frames else (a.height, a.width)
by = (18 * a.width * a.height + 18 * fh * fw) * a.frames * max(1, a.scenes)
res = []
for rnd in range(2):
    for m in ("31", "0", "3", "5", "7"):
        os.environ["CAMA_OVERLAY_CHUNK_LOG2"] = m
        for _ in range(4): job.step()
        job.eng.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): job.step()
        job.eng.join(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        res.append("%s:%.3f" % (m, by / dt / 8e12))
print(" ".join(argv), "|", " ".join(res))
