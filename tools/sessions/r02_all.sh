#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/sessions/r02_verify.sh
bash tools/sessions/r02_final.sh
