#!/bin/bash
# one steady-state step (100 000 regions, one chain) kernel by kernel: start and end of every launch relative to the step's first
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/run -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions ${1:-100000} --steps 3 --warmup 1 --in-flight 1 --sub-batches 1 > $OUT/log 2>&1
f=$(find $OUT/run -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last k_prepack_fast on
idx = [i for i, r in enumerate(rows) if "k_prepack" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f us  grid %8s lds %6s scratch %5s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""), r["Kernel_Name"][:70]))
PY
rm -rf $OUT/run
