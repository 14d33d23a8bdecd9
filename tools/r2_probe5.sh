#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r2m; rm -rf $OUT; mkdir -p $OUT
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu --no-e2e $EXTRA > $OUT/$name.json 2>> $OUT/err; }
EXTRA="" run c2_pair IHP_KSW_PAIR=1
EXTRA="--no-check" run c2_single IHP_KSW_PAIR=0
EXTRA="--no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1" run t100k_pair IHP_KSW_PAIR=1
EXTRA="--no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1" run t100k_single IHP_KSW_PAIR=0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2m/*.json")):
    try:
        d=json.load(open(f)); print("%-30s %10.0f %7.3f %s %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"], d.get("oracle_check",{}).get("identical")))
    except Exception as e: print(f, "ERR", e)
PY
