#!/bin/bash
# kernel stats of the steady-state launch (100 000 regions, one chain): rocprofv3 --kernel-trace --stats, the per-kernel averages
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r6s; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1 "$@" > $OUT/log 2>&1
f=$(find $OUT/run -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r6s/kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $OUT/run
