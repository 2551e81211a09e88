# decoder-alone rate of two library builds on one box (alternating): tools/ab_jpeg.sh <old.so> <new.so>
for rep in 1 2; do for lib in $1 $2; do
  echo "== $lib"; CAMA_ALLOW_LIB_OVERRIDE=1 CAMA_HIP_LIB=$PWD/$lib python tools/jpeg_probe.py --batch 240 --reps 5 2>&1 | grep "images/s =" 
done; done
