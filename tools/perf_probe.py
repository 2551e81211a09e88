"""Quick kernel-level timing probe (not the bench): main config, HIP events around render_frames."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cama_amd.engine import Engine
from tests.test_gpu_kernels import _random_scene, _rig

def main():
    N = int(os.environ.get("N", 10000)); F = int(os.environ.get("F", 40)); W = int(os.environ.get("W", 1600)); H = int(os.environ.get("H", 900))
    e = Engine("cuda:0")
    xyz, col, cams, w2c = _random_scene(1, N, F, W, H, spread=25.0)
    rig = _rig(e, cams); dmap = e.upload_map(xyz, col)
    src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty(e.mosaic_shape(rig, F), dtype=torch.uint8, device="cuda")
    T = e._mats(w2c)
    for _ in range(3): e.render_frames(dmap, rig, T, src, out)
    torch.cuda.synchronize()
    vu, vis, _ = e.project_frames(dmap, rig, T)
    print("visible stamps/frame", vis.sum().item() / F)
    reps = 10
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): e.render_frames(dmap, rig, T, src, out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    B = 13 * N + 36 * W * H
    print(f"N={N} F={F} {W}x{H}: {ms:.3f} ms/step  {F/ms*1e3:.0f} fps  {B*F/ms/1e6:.1f} GB/s ({B*F/ms/1e6/8000*100:.1f}% of 8 TB/s)")
    # plain copy ceiling for reference
    a = src.view(-1); b = torch.empty_like(a)
    for _ in range(2): b.copy_(a)
    t0.record()
    for _ in range(reps): b.copy_(a)
    t1.record(); torch.cuda.synchronize()
    ms2 = t0.elapsed_time(t1) / reps
    print(f"torch copy of src: {ms2:.3f} ms  {2*a.numel()/ms2/1e6:.1f} GB/s")
main()
