#!/bin/bash
# k_asm_combine3 time against the regions a CU holds (comb_occ / comb_minw knobs), steady 100 000-region launches under rocprofv3.
#   tools/r5_occ.sh "<knobs of run 1>" "<knobs of run 2>" ...     e.g.  tools/r5_occ.sh "comb_occ=16" "comb_occ=22 comb_minw=6"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/ks; rm -rf $OUT; mkdir -p $OUT
i=0
for kn in "$@"; do
  i=$((i+1)); name=run$i; args=""; for k in $kn; do args="$args --knob $k"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 2 --in-flight 1 --sub-batches 1 --knob verbose=1 $args > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  echo "== [$kn]: $(grep '^{"metric' $OUT/$name.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null)"
  grep "first tier" $OUT/$name.log | tail -1 | cut -c1-150
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_asm_combine3<" in r["Name"] and int(r["Calls"]) > 5: print("  ", r["Name"][10:42], r["Calls"], "launches, avg ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
  rm -rf $OUT/$name
done
