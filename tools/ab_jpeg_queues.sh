# decoder-alone rate over library builds x HW queue counts on one box (alternating):
#   tools/ab_jpeg_queues.sh "<lib.so> ..." "<GPU_MAX_HW_QUEUES values>" [jpeg_probe.py arguments]
libs=$1; qs=$2; shift 2
for rep in 1 2; do for lib in $libs; do for q in $qs; do
  echo "== $lib queues=$q $*"; GPU_MAX_HW_QUEUES=$q CAMA_ALLOW_LIB_OVERRIDE=1 CAMA_HIP_LIB=$PWD/$lib python tools/jpeg_probe.py --batch 240 --reps 5 "$@" 2>&1 | grep "images/s ="
done; done; done
