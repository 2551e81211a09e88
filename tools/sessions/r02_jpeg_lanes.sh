#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for l in 0 4 10 16 24; do echo "DRI lanes $l: $(timeout 300 python tools/jpeg_probe.py --batch 240 --lanes $l --min-group 4 --restart-rows 1 2>&1 | grep '^photo: 315' | cut -c1-75)"; done
