#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
for n in 4 8; do
CAMA_BENCH_SHARE_GPU=1 CAMA_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 6 --warmup 1 > $O/r02_bench_${n}ranks_shared_gpu.json 2> $O/r02_bench_${n}ranks.err; echo "n=$n rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/r02_bench_${n}ranks_shared_gpu.json") if l.startswith("{")][0])
print(d["n_gpus"], round(d["value"]), d["scaling"], d["hash_check"]["verified"], d["hash_check"]["mismatched"], "stress", round(d["stress"]["value"]), d["stress"]["hash_check"]["verified"], d.get("rccl_world"))
PY
done
