#!/bin/bash
mkdir -p gpurun_out/prof2
IHP_PROFILE=1 timeout 300 python bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 2 --warmup 1 --sub-batches 1 > gpurun_out/prof2/steady_prof.json 2> gpurun_out/prof2/err
python - <<PY
import json
d=json.load(open("gpurun_out/prof2/steady_prof.json")); print(d["kernel_ms"]); print(d.get("profile_cycles"))
PY
