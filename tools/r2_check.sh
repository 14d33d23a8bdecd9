#!/bin/bash
# quick GPU check: parity tests + default bench + 100k steady bench; outputs under gpurun_out/$1
T=${1:-chk}
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; tail -5 gpurun_out/$T/pytest.log
timeout 300 python bench.py --no-cpu --no-e2e > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
timeout 300 python bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1 > gpurun_out/$T/steady.json 2>> gpurun_out/$T/bench.err
python - <<PY
import json
for f in ("bench","steady"):
    try:
        d=json.load(open("gpurun_out/$T/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["kernel_ms"], d.get("oracle_check",{}).get("identical"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/$T/bench.err
