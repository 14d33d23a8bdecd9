#!/bin/bash
# instruction-cache counters of the ksw2 sweeps on synthetic jobs (is a 111 KB kernel with 32 waves per CU in different phases
# fetch-bound?)   tools/r5_icache.sh [jobs]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/ic; mkdir -p gpurun_out/ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic/g1 -- python3 tools/ksw_pair_time.py ${1:-20000} 247 identical > gpurun_out/ic/g1.log 2>&1
python3 tools/pmc_sum.py gpurun_out/ic/g1 --last 1 --json gpurun_out/ic/ic.json > gpurun_out/ic/sum.txt 2>&1
python3 - <<PY
import json
try:
    m=json.load(open("gpurun_out/ic/ic.json"))
    for k,v in m.items():
        if "k_ksw" in k and "plan" not in k:
            print(k, {x: round(y) for x,y in v.items() if isinstance(y,(int,float)) and not x.startswith("_")})
except Exception as e:
    print("failed", e); print(open("gpurun_out/ic/g1.log").read()[-1500:])
PY
rm -rf gpurun_out/ic/g1
