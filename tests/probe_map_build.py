"""Per-clip static-map build (reproject.py:72-106) for a ~1e6-point CAMA label set: reference-structured Python loops
(oracle), the product's vectorised host build, and the device kernel (cama_build_static_map)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cama_amd.reproject import MapManager
from cama_amd import runtime
from oracle import cama_oracle as O

rng = np.random.default_rng(0)
n_lines = int(os.environ.get("LINES", 2000))
labels = []
for i in range(n_lines):
    a0 = rng.uniform(100, 2800, 2)
    d = rng.normal(0, 1, 2); d /= np.linalg.norm(d)
    t = np.linspace(0, 50.0, 11)[:, None]
    labels.append({"attrs": {"type": ["lane_marking", "Road_teeth", "Crosswalk_Line"][i % 3]}, "data": (a0 + t * d).tolist()})
bev = rng.normal(0, 0.05, (3000, 3000)).astype(np.float32)
mm = MapManager()
t0 = time.perf_counter(); host = mm.calculate_3d_instance_maps(bev, labels); t1 = time.perf_counter()
N = sum(p["points"].shape[0] for p in host)
print(f"N = {N} densified points from {n_lines} labels")
print(f"vectorised host build (product default): {(t1 - t0) * 1e3:.1f} ms")
sub = labels[: max(1, n_lines // 20)]
t0 = time.perf_counter(); O.static_map_cama(bev, sub); t1 = time.perf_counter()
print(f"reference-structured Python loops (oracle), extrapolated from {len(sub)} labels: {(t1 - t0) * n_lines / len(sub):.1f} s")
eng = runtime.engine()
t0 = time.perf_counter(); table = mm.segment_table(labels); t1 = time.perf_counter()
print(f"segment table on host: {(t1 - t0) * 1e3:.1f} ms")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dmap = eng.build_static_map(table, lift=True, bev_height=bev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"device build incl. uploads (36 MB raster): {(t1 - t0) * 1e3:.2f} ms")
from cama_amd import _lib
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
raster = torch.from_numpy(bev).cuda()
up = {k: torch.from_numpy(np.ascontiguousarray(table[k])).cuda() for k in ("verts", "seg_v0", "seg_num", "seg_off", "seg_colour")}
soa = torch.empty((3, N), dtype=torch.float32, device="cuda"); col = torch.empty(N, dtype=torch.uint8, device="cuda")
L = _lib.lib()
def launch():
    _lib.check(L.cama_build_static_map(up["verts"].data_ptr(), up["seg_v0"].data_ptr(), up["seg_num"].data_ptr(), up["seg_off"].data_ptr(),
               up["seg_colour"].data_ptr(), len(table["seg_num"]), N, 1, raster.data_ptr(), 0, 3000, 3000, 0.1, 300.0, 300.0, 0.0, 0.0,
               soa.data_ptr(), soa.data_ptr() + 4 * N, soa.data_ptr() + 8 * N, col.data_ptr(), torch.cuda.current_stream().cuda_stream))
launch(); torch.cuda.synchronize()
ev0.record()
for _ in range(20): launch()
ev1.record(); torch.cuda.synchronize()
print(f"kernel alone: {ev0.elapsed_time(ev1) / 20 * 1e3:.1f} us  ({13 * N / (ev0.elapsed_time(ev1) / 20 * 1e-3) / 1e9:.0f} GB/s of output)")
assert np.array_equal(dmap.soa.cpu().numpy().T, np.concatenate([p["points"] for p in host]))
print("device == host, bit for bit")
