# occupancy sweep of the global-words decoder (LDS padding), photo + noise, alternating; then per-kernel stats old vs new
for rep in 1 2; do
for lib in tools/_abtmp/libcama_r5.so cama_amd/libcama_hip.so tools/_abtmp/libcama_pad6144.so tools/_abtmp/libcama_pad12288.so tools/_abtmp/libcama_pad20480.so tools/_abtmp/libcama_pad33792.so; do
  echo "== $lib"
  CAMA_HIP_LIB=$PWD/$lib timeout 300 python tools/jpeg_probe.py --batch 240 --reps 5 2>&1 | grep "images/s ="
done; done
export TMPDIR=/tmp
R=$PWD
for lib in tools/_abtmp/libcama_r5.so cama_amd/libcama_hip.so tools/_abtmp/libcama_pad20480.so; do
  echo "== kernel stats (photo only) $lib"
  o=$R/gpurun_out/jpeg_stats_$(basename $lib .so)
  (cd /tmp && CAMA_HIP_LIB=$R/$lib timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $o -o j -- python $R/tools/jpeg_probe.py --batch 240 --reps 5 --sets photo > $o.log 2>&1)
  python - $o <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:11]:
    print("%-50s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  grep "images/s =" $o.log
done
