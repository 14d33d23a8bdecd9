#!/bin/bash
# Round-4 loop on the GPU box (one gpurun call): the pair sweep against the oracle, then pair / single bench lines on the same box.
#   tools/r4_check.sh [seed] [batches]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 800 python tools/ksw_pair_stress.py ${1:-5} ${2:-100} > gpurun_out/pair_stress.txt 2>&1
grep DIFF -A4 gpurun_out/pair_stress.txt | head -60; tail -2 gpurun_out/pair_stress.txt
: > gpurun_out/b.err
python bench.py --no-cpu --no-e2e --no-other > gpurun_out/b_pair.json 2>> gpurun_out/b.err
python bench.py --no-cpu --no-e2e --no-other --no-check --knob ksw_pair=0 > gpurun_out/b_single.json 2>> gpurun_out/b.err
python bench.py --no-cpu --no-e2e --no-other --no-check --regions 100000 --in-flight 1 --sub-batches 1 --steps 5 > gpurun_out/b_pair100k.json 2>> gpurun_out/b.err
python bench.py --no-cpu --no-e2e --no-other --no-check --regions 100000 --in-flight 1 --sub-batches 1 --steps 5 --knob ksw_pair=0 > gpurun_out/b_single100k.json 2>> gpurun_out/b.err
grep -v amdgpu.ids gpurun_out/b.err | tail -5
for f in b_pair b_single b_pair100k b_single100k; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), d["ms_per_step"], d.get("kernel_ms"), (d.get("oracle_check") or {}).get("identical"))
except Exception as e:
    print("$f", "failed", e)
PY
done
