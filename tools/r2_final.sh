#!/bin/bash
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; tail -3 gpurun_out/final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; python - <<PY
import json
d=json.load(open("gpurun_out/final/bench.json")); print(d["value"], d["ms_per_step"], d["kernel_ms"], d["oracle_check"]["identical"], d["roofline"]["frac"], d["roofline"]["issue"]["issue_frac"], d["cpu_baseline"]["value"])
PY
