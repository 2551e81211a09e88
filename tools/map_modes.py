#!/usr/bin/env python3
"""Workgroup -> band orders of the overlay compared INSIDE one process (the contiguous order's speed differs between
processes, profiles/r03_process_modes.txt, so comparisons across processes say little).  Needs an A/B build of the library
that re-reads the knob at every launch:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -DOVERLAY_MAP_SWITCH \\
          cama_amd/csrc/cama_hip.hip -o tools/ab/libcama_mapswitch.so
    CAMA_HIP_LIB=$PWD/tools/ab/libcama_mapswitch.so python tools/map_modes.py [bench.py workload arguments]

Prints, for two rounds, whole step / 8 TB/s under CAMA_OVERLAY_CHUNK_LOG2 = 31 (contiguous per XCD), 0 (interleaved) and
round-robin chunks of 8 / 32 / 128 bands."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

argv = sys.argv[1:]
a = bench.parse_args(argv)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
job = bench.Job(a, list(range(max(1, a.scenes))), dev, None)
fh, fw = (900, 1600) if a.raw_frames else (a.height, a.width)
by = (18 * a.width * a.height + 18 * fh * fw) * a.frames * max(1, a.scenes)
res = []
for rnd in range(2):
    for m in ("31", "0", "3", "5", "7"):
        os.environ["CAMA_OVERLAY_CHUNK_LOG2"] = m
        for _ in range(4):
            job.step()
        job.eng.join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            job.step()
        job.eng.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        res.append("%s:%.3f" % (m, by / dt / 8e12))
print(" ".join(argv), "|", " ".join(res))
