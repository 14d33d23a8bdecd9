#!/bin/bash
# Round-5 loop on the GPU box (one gpurun call): the GPU suite (or a subset), then a default bench line.
#   tools/r5_check.sh [pytest -k expression | all | none] [bench args...]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
sel=${1:-all}; shift
if [ "$sel" = all ]; then timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest.txt 2>&1
elif [ "$sel" != none ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$sel" > gpurun_out/r5_pytest.txt 2>&1; fi
tail -5 gpurun_out/r5_pytest.txt 2>/dev/null
t0=$(date +%s); timeout 1500 python bench.py "$@" > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; echo "bench wall $(( $(date +%s) - t0 )) s"
grep -v amdgpu.ids gpurun_out/r5_bench.err | tail -5
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_bench.json").read().strip().splitlines()[-1])
    print(json.dumps(d["digest"]))
except Exception as e:
    print("bench failed", e)
PY
