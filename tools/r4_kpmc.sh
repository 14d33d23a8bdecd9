#!/bin/bash
# instruction counts of the ksw2 sweeps on synthetic jobs (rocprofv3 --pmc, one scenario)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/kp; mkdir -p gpurun_out/kp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d gpurun_out/kp/g1 -- python3 tools/ksw_pair_time.py ${1:-100000} ${2:-247} ${3:-identical} ${4:-50} > gpurun_out/kp/g1.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/kp/g2 -- python3 tools/ksw_pair_time.py ${1:-100000} ${2:-247} ${3:-identical} ${4:-50} > gpurun_out/kp/g2.log 2>&1
python3 tools/pmc_sum.py gpurun_out/kp/g1 gpurun_out/kp/g2 --last 1 --json gpurun_out/kp/mix.json > /dev/null 2>&1
python3 - <<PY
import json
m=json.load(open("gpurun_out/kp/mix.json"))
for k,v in m.items():
    if "k_ksw" in k and "plan" not in k:
        print(k, {x: round(y) for x,y in v.items() if isinstance(y,(int,float)) and not x.startswith("_")})
PY
rm -rf gpurun_out/kp/g1 gpurun_out/kp/g2
