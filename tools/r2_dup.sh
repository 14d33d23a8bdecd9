#!/bin/bash
mkdir -p gpurun_out/dup
timeout 600 python -m pytest tests -x -q -m gpu -k "fallback or dup or golden or sweep" > gpurun_out/dup/pytest.log 2>&1; tail -2 gpurun_out/dup/pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu --no-e2e --no-check --dup-frac 0.1 > gpurun_out/dup/dup$i.json 2>> gpurun_out/dup/err
python - <<PY
import json
d=json.load(open("gpurun_out/dup/dup$i.json")); print("dup", d["value"], d["ms_per_step"], d["kernel_ms"])
PY
done
