#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out
for g in 300 330 350 370 380 384 400 440 500; do echo "group_wgs $g: $(CAMA_JPEG_GROUP_WGS=$g timeout 300 python tools/jpeg_probe.py --batch 240 2>&1 | grep '^photo: 315' | cut -c1-75)"; done
echo "equal 384: $(CAMA_JPEG_SPLIT=equal timeout 300 python tools/jpeg_probe.py --batch 240 2>&1 | grep '^photo: 315' | cut -c1-75)"
timeout 300 python tools/jpeg_probe.py --batch 240 2>&1 | grep '^noise: 1298' | cut -c1-75
timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q 2>&1 | tail -2
