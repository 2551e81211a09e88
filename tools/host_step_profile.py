"""cProfile of the host side of one bench step (ClipManager.render_clip, pipelined) at 960x540, where the step is host-bound."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse_args(["--height", "540", "--width", "960", "--cpu-seconds", "0"])
dev = torch.device("cuda:0")
job = bench.Job(a, [0], dev)
for _ in range(50): job.step()
job.eng.join(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(2000): job.step()
t1 = time.perf_counter()
job.eng.join(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue: {(t1 - t) / 2000 * 1e6:.1f} us per step; with the GPU: {(t2 - t) / 2000 * 1e6:.1f} us per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): job.step()
pr.disable()
job.eng.join()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
