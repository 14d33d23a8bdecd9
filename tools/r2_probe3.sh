#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r2i; rm -rf $OUT; mkdir -p $OUT
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
IHP_PROFILE=1 timeout 300 python3 bench.py --no-cpu --no-e2e --sub-batches 1 > $OUT/p.json 2>> $OUT/err
timeout 300 python3 bench.py --no-cpu --no-e2e --no-check > $OUT/b.json 2>> $OUT/err
timeout 300 python3 bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1 > $OUT/t_100k.json 2>> $OUT/err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 bench.py --no-cpu --no-e2e --no-check --steps 5 --warmup 2 > $OUT/c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t100k -- python3 bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1 > $OUT/t100k.log 2>&1
for d in t100k c2; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; done
python3 - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("gpurun_out/r2i/*.json")):
    try:
        d=json.load(open(f)); p=d.get("profile_cycles"); print(f, d["value"], d["ms_per_step"], d["kernel_ms"], d.get("oracle_check",{}).get("identical"), {i:v for i,v in enumerate(p) if v} if p else "")
    except Exception as e: print(f, "ERR", e)
for n in ("t100k","c2"):
    print("==", n)
    for row in list(csv.DictReader(open("gpurun_out/r2i/%s_kernel_stats.csv"%n)))[:14]:
        print("  %-60s calls %4s avg %10.1f us  %5s%%" % (row["Name"][:60], row["Calls"], float(row["AverageNs"])/1e3, row["Percentage"]))
PY
