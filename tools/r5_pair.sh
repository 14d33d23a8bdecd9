#!/bin/bash
# Round-5 loop for the pair sweep (one gpurun call): stress against the oracle, the reference-generated groups, kernel time and
# instruction counts on synthetic jobs, then a C2 bench line.
#   tools/r5_pair.sh [seed] [batches] [jobs for the timing]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python tools/ksw_pair_stress.py ${1:-5} ${2:-60} > gpurun_out/pair_stress.txt 2>&1
grep DIFF -A4 gpurun_out/pair_stress.txt | head -40; tail -2 gpurun_out/pair_stress.txt
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -m gpu -x -q -k "pair or paired or sweep" 2>&1 | tail -3
tools/r4_ktime.sh ${3:-100000} 247 2>&1 | tail -12
tools/r4_kpmc.sh 50000 247 identical 2>&1 | tail -4
python bench.py --no-cpu --no-e2e --no-other > gpurun_out/b_pair.json 2> gpurun_out/b.err
python bench.py --no-cpu --no-e2e --no-other --no-check --regions 100000 --in-flight 1 --sub-batches 1 --steps 5 > gpurun_out/b_pair100k.json 2>> gpurun_out/b.err
grep -v amdgpu.ids gpurun_out/b.err | tail -5
for f in b_pair b_pair100k; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), d["ms_per_step"], d.get("kernel_ms"), (d.get("oracle_check") or {}).get("identical"))
except Exception as e:
    print("$f", "failed", e)
PY
done
