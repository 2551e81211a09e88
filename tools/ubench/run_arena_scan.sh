#!/bin/bash
# one 50 GB allocation carved into 24 (frames, mosaic) sets at fixed offsets: does the speed follow the OFFSET?
set -u
cd tools/ubench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include overlay_modes.cpp -o overlay_modes -L ../../cama_amd -lcama_hip -Wl,-rpath,'$ORIGIN/../../cama_amd' || exit 1
for p in 1 2; do REPS=8 ./overlay_modes arena 40 24 "$(cat arena_scan_script.txt)"; done > ../../gpurun_out/arena_scan.txt 2>&1
for p in 1; do REPS=8 ./overlay_modes contig 40 24 "$(cat arena_scan_script.txt)"; done > ../../gpurun_out/contig_scan.txt 2>&1
python3 - <<'PY'
import re
for name in ("arena_scan", "contig_scan"):
    for blk in open(f"../../gpurun_out/{name}.txt").read().split("#")[1:]:
        rows = re.findall(r"src (\d+) dst (\d+)\s+order\s+(\d+).*?med ([\d.]+)", blk)
        v = [int(float(m) * 1000) for s, d, o, m in rows]
        print(name, "dst scan (src 0):", v[:24]); print(name, "src scan (dst 0):", v[24:48]); print(name, "pair k,k        :", v[48:72])
PY
