# Is the GPU busy while batches of 240 photo-like JPEGs are decoded back to back?  rocprofv3 kernel + copy trace of
# tools/jpeg_concurrency_probe.py (1, 2, 3 batches in flight), then per concurrency level: wall, union of kernel intervals,
# sum of kernel durations (= mean number of kernels running), union of copy intervals, time per kernel type.
#   tools/jpeg_busy.sh [tag]
export TMPDIR=/tmp
R=$PWD
tag=${1:-busy}
o=$R/gpurun_out/jpeg_$tag
(cd /tmp && PROBE_GROUPS=0 timeout 600 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $o -o j -- python $R/tools/jpeg_concurrency_probe.py > $o.log 2>&1)
grep "in flight" $o.log
python - $o <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
K = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
C = []
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
for r in csv.DictReader(open(mc[0])) if mc else []:
    C.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?")))
K.sort(); C.sort()
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
# windows: split the trace at idle gaps > 300 us between jpeg kernels; keep the long windows (the timed loops of 5 repetitions)
J = [k for k in K if k[2].startswith("k_jpeg") or "fillBuffer" in k[2]]
wins, cur = [], [J[0]]
for k in J[1:]:
    if k[0] - max(x[1] for x in cur[-40:]) > 300000:
        wins.append(cur); cur = [k]
    else:
        cur.append(k)
wins.append(cur)
for w in wins:
    t0, t1 = w[0][0], max(x[1] for x in w)
    if t1 - t0 < 8e6: continue                      # (warm-ups)
    wall = (t1 - t0) / 1e3
    busy = union([(s, e) for s, e, _ in w]) / 1e3
    tot = sum(e - s for s, e, _ in w) / 1e3
    cop = [(s, e) for s, e, _ in C if s >= t0 and e <= t1]
    n_count = sum(1 for x in w if x[2] == "k_jpeg_count")
    per = collections.Counter()
    for s, e, n in w: per[n] += (e - s) / 1e3
    print("window %.1f ms, %d group chains: kernels busy %.1f ms (%.0f %%), sum of kernel durations %.1f ms (mean %.2f running), copies busy %.1f ms (%.0f %%)"
          % (wall / 1e3, n_count, busy / 1e3, 100 * busy / wall, tot / 1e3, tot / busy, union(cop) / 1e6, 100 * union(cop) / 1e3 / wall))
    print("   per chain, us: " + "  ".join("%s %.0f" % (n.replace("k_jpeg_", ""), v / max(1, n_count)) for n, v in per.most_common()))
PY
