"""cProfile of the host side of one pipelined step (render_clip) -- where the ~150 us per step go when the GPU step
is shorter than that (960x540).  Usage: python tools/host_profile.py [--height 540 --width 960]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=540)
ap.add_argument("--width", type=int, default=960)
a = ap.parse_args()
ns = argparse.Namespace(frames=40, verts=10000, height=a.height, width=a.width, map="lanes")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cm, frames, clip = bench.build_scene(ns, 0, dev)
from cama_amd import runtime
eng = runtime.engine(); rig = cm._rig()
out = torch.empty(eng.mosaic_shape(rig, 40), dtype=torch.uint8, device=dev)
for _ in range(20): cm.render_clip("cama", out=out, pipelined=True)
eng.join(); torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): cm.render_clip("cama", out=out, pipelined=True)
t1 = time.perf_counter(); eng.join(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"issue {(t1-t0)/n*1e6:.1f} us/step, incl. drain {(t2-t0)/n*1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): cm.render_clip("cama", out=out, pipelined=True)
pr.disable(); eng.join(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
