# decoder-alone rate of two TREES on one box (alternating) -- for changes that touch the host side or the ABI:
#   tools/ab_jpeg_trees.sh <dirA> <dirB> [jpeg_probe.py arguments]     (a tree = a checkout with its built libcama_hip.so)
A=$1; B=$2; shift 2
for rep in 1 2 3; do for d in $A $B; do
  echo "== $d $*"; (cd $d && python tools/jpeg_probe.py --batch 240 --reps 5 "$@" 2>&1 | grep "images/s =")
done; done
