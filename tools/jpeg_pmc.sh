# SQ counters of the JPEG entropy kernels on a batch that FILLS the chip (240 photo-like images, 7 groups): tools/jpeg_pmc.sh [tag]
# (counter passes only: --kernel-trace + --pmc, one small set per process)
export TMPDIR=/tmp
R=$PWD
tag=${1:-jpmc}
o=$R/gpurun_out/$tag
mkdir -p $o
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $o/p$i -o c -- python $R/tools/jpeg_probe.py --batch 240 --reps 1 --sets ${JPEG_SETS:-photo} > $o/p$i.log 2>&1)
done
python - $o <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "jpeg" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print(f"    {c:26s} mean per dispatch {tot[k][c] / n[k][c]:16.1f}   ({n[k][c]} dispatches)")
PY
