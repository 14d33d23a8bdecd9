#!/bin/bash
mkdir -p gpurun_out/c5d
run() { tag=$1; shift; env "$@" IHP_PROFILE=1 timeout 300 python bench.py --config C5 --no-cpu --no-e2e --no-check --steps 3 --warmup 1 > gpurun_out/c5d/$tag.json 2>> gpurun_out/c5d/err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/c5d/$tag.json")); p=d["profile_cycles"]; print("$tag", d["value"], d["ms_per_step"], d["kernel_ms"], "handed_back", p[23], "pass2/3/4", p[24],p[25],p[26], "roomy", p[28], "contigs", d["results"]["contigs"])
except Exception as e: print("$tag failed", e)
PY
}
run base X=1
run occ8 IHP_V2_OCC=8
run occ6 IHP_V2_OCC=6
run occ10 IHP_V2_OCC=10
run v1 IHP_ASM_V1=1
