# kernel timeline of ONE decoded batch (240 photo-like images): tools/jpeg_timeline.sh [tag] [jpeg_probe.py arguments]
export TMPDIR=/tmp
R=$PWD
tag=${1:-default}; shift
o=$R/gpurun_out/jpeg_tl_$tag
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $o -o j -- python $R/tools/jpeg_probe.py --batch 240 --reps 3 "$@" > $o.log 2>&1)
python - $o <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:22], r["Queue_Id"]))
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
for r in csv.DictReader(open(mc[0])) if mc else []:
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")[-14:], "-"))
rows.sort()
# the last batch = everything after the last k_jpeg_count that follows a gap: take the last 7 groups' kernels
cnt = [i for i, r in enumerate(rows) if r[2].startswith("k_jpeg_count")]
# a batch starts at a count kernel before which every earlier group has finished (as many colour kernels ended as count
# kernels started): decode() returns only when all its groups are done
col_end = sorted(r[1] for r in rows if r[2].startswith("k_jpeg_colour"))
import bisect
first = cnt[0]
for rank, i in enumerate(cnt):
    if bisect.bisect_right(col_end, rows[i][0]) == rank:
        first = i
# walk back over copies just before
t0 = rows[first][0]
sel = [r for r in rows[first:] ]
print("last batch: %d records, wall %.1f us" % (len(sel), (max(r[1] for r in sel) - t0) / 1e3))
per = collections.OrderedDict()
for s, e, n, q in sel:
    per.setdefault(q, []).append((n, (s - t0) / 1e3, (e - t0) / 1e3))
for q, ks in per.items():
    print("queue", q, " ".join("%s[%.0f-%.0f]" % (n.replace("k_jpeg_", ""), a, b) for n, a, b in ks))
PY
grep "images/s =" $o.log | tail -2
