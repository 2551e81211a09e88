#!/bin/bash
# 128 frames per launch (3.3 GB read + 3.3 GB written): 12 destinations for source 0, 12 sources for destination 0.
set -u
cd tools/ubench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include overlay_modes.cpp -o overlay_modes -L ../../cama_amd -lcama_hip -Wl,-rpath,'$ORIGIN/../../cama_amd' || exit 1
for p in 1 2; do REPS=8 ./overlay_modes malloc 128 12 "$(cat scan128_script.txt)"; done > ../../gpurun_out/scan128.txt 2>&1
cut -c1-60,100- ../../gpurun_out/scan128.txt | awk '{print $2,$4,$6,$16,$22}' | tr '\n' ';'
