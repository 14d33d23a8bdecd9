#!/bin/bash
# Per-kernel time (rocprofv3 kernel stats) of the steady 100 000-region launch, C5 and C3 -- the quick look between two edits.
#   [KNOBS="--knob k=v ..."] tools/r5_kstats.sh [steady|c5|c3 ...]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/ks; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py "$@" $KNOBS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); echo "== $name: $(tail -1 $OUT/$name.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ihp::" in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:14]:
    print("  %-52s %4s launches  avg %9.3f us  total %8.2f ms" % (r["Name"].replace("void ", "").replace("ihp::", "")[:52], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $OUT/$name; }
for w in ${@:-steady c5 c3}; do
  case $w in
    steady) run steady --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1;;
    c5) run c5 --config C5 --steps 5 --warmup 2 --no-cpu --no-e2e --no-check;;
    c3) run c3 --config C3 --steps 3 --warmup 1 --no-cpu --no-e2e --no-check;;
  esac
done
