#!/bin/bash
mkdir -p gpurun_out/sweep
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --no-e2e --no-check > gpurun_out/sweep/$tag.json 2>> gpurun_out/sweep/err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sweep/$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["kernel_ms"])
except Exception as e: print("$tag failed", e)
PY
}
run base X=1
run occ20 IHP_V2_OCC=20
run occ24 IHP_V2_OCC=24
run occ18 IHP_V2_OCC=18
run base2 X=1
