#!/bin/bash
# the N = 4 and N = 8 code paths of bench.py on ONE GPU (ranks share it, gloo): sharding, stress ranges, hash checks
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
for n in 8; do
  CAMA_BENCH_SHARE_GPU=1 CAMA_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 4 --warmup 1 > $O/r02v_${n}ranks.json 2> $O/r02v_${n}ranks.err; echo "n=$n rc=$?"
  python - "$O/r02v_${n}ranks.json" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "n_gpus", "scaling", "rccl_world")}, d["hash_check"]["verified"], d["hash_check"]["mismatched"], d["hash_check"]["missing"], [round(x) for x in d["per_rank_frames"]])
        print("  stress", round(d["stress"]["value"]), d["stress"]["hash_check"]["verified"], d["stress"]["hash_check"]["missing"], [round(x) for x in d["stress"]["per_rank_frames"]])
PY
  tail -2 $O/r02v_${n}ranks.err | cut -c1-200
done
