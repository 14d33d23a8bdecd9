"""Diagnostics: merged timeline of kernels and memory copies from a rocprofv3 --kernel-trace --memory-copy-trace run (csv)."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("ihp::")[-1][:26], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
last = ev[-n:]
t0 = last[0][0]
for s, e, name, q in last:
    print("%9.3f -> %9.3f ms  dur %7.3f  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, name, q))
