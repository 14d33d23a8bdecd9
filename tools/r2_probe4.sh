#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r2j; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu --no-e2e --no-check $EXTRA > $OUT/$name.json 2>> $OUT/err; }
for S in 2 3 4; do
  EXTRA="--sub-batches $S" run pm1_occ20_s$S IHP_V2_PM=1 IHP_V2_OCC=20
  EXTRA="--sub-batches $S" run pm0_occ20_s$S IHP_V2_PM=0 IHP_V2_OCC=20
  EXTRA="--sub-batches $S" run pm1_occ16_s$S IHP_V2_PM=1 IHP_V2_OCC=16
  EXTRA="--sub-batches $S" run pm0_occ16_s$S IHP_V2_PM=0 IHP_V2_OCC=16
done
for W in 16 24 32; do
  EXTRA="--regions 100000 --steps 3 --warmup 1 --sub-batches 1" run t100k_ksw$W IHP_KSW_WAVES=$W
done
EXTRA="--regions 100000 --steps 3 --warmup 1 --sub-batches 1" run t100k_pm0 IHP_V2_PM=0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2j/*.json")):
    try:
        d=json.load(open(f)); print("%-40s %10.0f %7.3f %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
