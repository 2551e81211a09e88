# one box: byte-equality suite on the new library, then old / new decoder-alone rate x group sizing
timeout 900 python -m pytest tests/test_gpu_jpeg.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for lib in tools/_abtmp/libcama_r5.so cama_amd/libcama_hip.so; do
for g in 0 768 1024 2048; do
  echo "== $lib group_wgs=$g"
  CAMA_JPEG_GROUP_WGS=$g CAMA_HIP_LIB=$PWD/$lib timeout 300 python tools/jpeg_probe.py --batch 240 --reps 5 2>&1 | grep "images/s ="
done; done; done
