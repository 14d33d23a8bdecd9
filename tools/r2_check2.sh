#!/bin/bash
T=${1:-chk}
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
run() { tag=$1; shift; IHP_PROFILE=1 timeout 300 python bench.py --no-cpu --no-e2e "$@" > gpurun_out/$T/$tag.json 2>> gpurun_out/$T/err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$T/$tag.json")); p=d.get("profile_cycles",[0]*32); print("$tag", d["value"], d["ms_per_step"], d["kernel_ms"], d.get("oracle_check",{}).get("identical"), "handed_back", p[23], "roomy", p[28])
except Exception as e: print("$tag failed", e)
PY
}
run c2
run steady --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1
run c5 --config C5 --steps 3 --warmup 1
run c3 --config C3 --steps 2 --warmup 1 --no-check
tail -3 gpurun_out/$T/err
