"""From a rocprofv3 --kernel-trace --memory-copy-trace run of tools/e2e_threads.py: how busy the GPU and the copy engines were in
the last stretch of the run, and the kernel time per batch.   python tools/e2e_busy.py <dir> [n_last_kernels]"""
import csv, glob, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 600
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(kf))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
kiv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
print("window %.2f ms, %d kernels; GPU busy (union of kernel intervals) %.3f; sum of kernel times / window %.3f" % ((t1 - t0) / 1e6, len(rows), union(kiv) / (t1 - t0), sum(e - s for s, e in kiv) / (t1 - t0)))
by = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1][:30]
    by.setdefault(n, [0, 0]); by[n][0] += 1; by[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
nb = by.get("k_summary", [1])[0]
print("batches in the window: %d -> %.3f ms per batch" % (nb, (t1 - t0) / 1e6 / max(nb, 1)))
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-30s %5d launches  %8.3f ms per batch" % (n, c, t / 1e6 / max(nb, 1)))
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mf:
    ms = [r for r in csv.DictReader(open(mf[0])) if t0 <= int(r["Start_Timestamp"]) <= t1]
    for dirn in sorted({r["Direction"] for r in ms}):
        iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ms if r["Direction"] == dirn]
        print("  copies %-24s %4d  busy %.3f  %.3f ms per batch" % (dirn, len(iv), union(iv) / (t1 - t0), sum(e - s for s, e in iv) / 1e6 / max(nb, 1)))
