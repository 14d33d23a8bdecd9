#!/bin/bash
# k_asm_combine3 time against the regions a CU holds (comb_occ / comb_minw knobs), steady 100 000-region launches under rocprofv3.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/ks; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1 --knob verbose=1 "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); echo "== $name: $(grep '^{"metric' $OUT/$name.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null)"
  grep "first tier" $OUT/$name.log | tail -1 | cut -c1-120
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_asm_combine3<" in r["Name"] and int(r["Calls"])>5: print("  ", r["Name"][:40], r["Calls"], "avg ms", round(float(r["AverageNs"])/1e6,3))
PY
  rm -rf $OUT/$name; }
run o12 --knob comb_occ=12
run o16 --knob comb_occ=16
run o18 --knob comb_occ=18
run o20 --knob comb_occ=20
run o24w6 --knob comb_minw=6 --knob comb_occ=24
run o28w7 --knob comb_minw=7 --knob comb_occ=28
