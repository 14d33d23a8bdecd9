#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per launch.

usage: pmc_sum.py DIR [DIR ...] [--json OUT] [--last N]
Kernel names are shortened to the function name + template arguments.  --last N: only the last N launches of every kernel
in each file count (the runs after the library has seen a batch: no empty retry launches, tiers cut for the workload).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = None
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != out]
    last = 0
    if "--last" in sys.argv:
        last = int(sys.argv[sys.argv.index("--last") + 1])
        args = [a for a in args if a != str(last)]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    meta = {}
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            if last:
                ids = defaultdict(set)
                for row in rows:
                    ids[short(row["Kernel_Name"])].add(int(row["Dispatch_Id"]))
                keep = {k: set(sorted(v)[-last:]) for k, v in ids.items()}
                rows = [row for row in rows if int(row["Dispatch_Id"]) in keep[short(row["Kernel_Name"])]]
            for row in rows:
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                launches[k][c].add((f, row["Dispatch_Id"]))
                meta[k] = {"vgpr": row.get("VGPR_Count"), "agpr": row.get("Accum_VGPR_Count"), "sgpr": row.get("SGPR_Count"),
                           "lds": row.get("LDS_Block_Size"), "scratch": row.get("Scratch_Size"), "grid": row.get("Grid_Size")}
    res = {}
    for k in sorted(acc):
        res[k] = {"_meta": meta[k]}
        for c in sorted(acc[k]):
            n = max(1, len(launches[k][c]))
            res[k][c] = acc[k][c] / n
            res[k]["_launches"] = n
    for k, v in res.items():
        print(k, v["_meta"], "launches", v.get("_launches"))
        for c, x in v.items():
            if not c.startswith("_"):
                print("   %-28s %16.1f" % (c, x))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
