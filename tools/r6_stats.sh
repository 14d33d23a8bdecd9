#!/bin/bash
# kernel stats of one bench configuration: tools/r6_stats.sh NAME [bench args...] -> gpurun_out/r6k/NAME_kernel_stats.csv (+ a short table on stdout)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/r6k; mkdir -p $OUT; rm -rf $OUT/run_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run_$NAME -- python3 bench.py --no-cpu --no-e2e --no-check --no-other "$@" > $OUT/$NAME.log 2>&1
f=$(find $OUT/run_$NAME -name "*kernel_stats.csv" | head -1)
cp $f $OUT/${NAME}_kernel_stats.csv
NAME=$NAME python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open("gpurun_out/r6k/%s_kernel_stats.csv" % os.environ["NAME"])))
for r in rows[:14]:
    print("%-62s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $OUT/run_$NAME
tail -1 $OUT/$NAME.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
