# per-kernel time of the device JPEG decoder (240 photo-like 1600x900 images per batch): tools/jpeg_kernel_stats.sh [lib.so]
export TMPDIR=/tmp
R=$PWD
o=$R/gpurun_out/jpeg_stats
[ -n "${1:-}" ] && export CAMA_ALLOW_LIB_OVERRIDE=1 CAMA_HIP_LIB=$R/$1
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $o -o j -- python $R/tools/jpeg_probe.py --batch 240 --reps 5 --sets ${JPEG_SETS:-noise,photo} > $o.log 2>&1)
python - $o <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "images/s =" $o.log
