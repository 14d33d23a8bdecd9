"""From a rocprofv3 --kernel-trace --memory-copy-trace run of tools/e2e_threads.py: kernels and copies of the last stretch in
start order (ms from the first one shown).   python tools/e2e_timeline.py <dir> [n_last]"""
import csv, glob, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = []
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:26], r.get("Queue_Id", "")))
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"].replace("MEMORY_COPY_", "")[:16], ""))
ev.sort()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # leave out the last `skip` events (the pipeline draining)
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None  # name filters
if skip:
    ev = ev[:-skip]
ev = ev[-n_last:]
if only:
    ev = [e for e in ev if any(o in e[2] for o in only)]
t0 = ev[0][0]
for s, e, n, q in ev:
    print("%9.3f -> %9.3f  %7.3f  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
