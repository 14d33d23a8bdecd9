#!/bin/bash
mkdir -p gpurun_out/sweep
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu --no-e2e --no-check "$@" > gpurun_out/sweep/$tag.json 2>> gpurun_out/sweep/err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sweep/$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["kernel_ms"])
except Exception as e: print("$tag failed", e)
PY
}
run sb1 --sub-batches 1
run sb2 --sub-batches 2
run sb3 --sub-batches 3
run sb4 --sub-batches 4
run sb2b --sub-batches 2
