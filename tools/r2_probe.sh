#!/bin/bash
# round-2 diagnostics: per-phase cycle counters and the latency / occupancy curve of k_assemble
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r2b; mkdir -p $OUT
IHP_PROFILE=1 python3 bench.py --no-cpu --sub-batches 1 --steps 5 > $OUT/prof_c2.json 2> $OUT/prof_c2.err
for W in 1 2 4 8 16; do
  IHP_ASM_WAVES=$W python3 bench.py --no-cpu --sub-batches 1 --regions $((256*W)) --steps 5 > $OUT/lat_w$W.json 2>> $OUT/lat.err
done
for W in 4 8 16; do
  IHP_ASM_WAVES=$W python3 bench.py --no-cpu --sub-batches 1 --regions 100000 --steps 3 --warmup 1 > $OUT/thr_w$W.json 2>> $OUT/lat.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2b/*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["kernel_ms"], d.get("profile_cycles"))
    except Exception as e: print(f, "ERR", e)
PY
