#!/bin/bash
# instruction mix of the kernels at steady state (100k regions, one chain)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r2k; rm -rf $OUT; mkdir -p $OUT
ARGS="--no-cpu --no-e2e --no-check --regions 100000 --steps 2 --warmup 1 --sub-batches 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/g1 -- python3 bench.py $ARGS > $OUT/g1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/g2 -- python3 bench.py $ARGS > $OUT/g2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $OUT/g3 -- python3 bench.py $ARGS > $OUT/g3.log 2>&1
python3 tools/pmc_sum.py $OUT/g1 $OUT/g2 $OUT/g3 --json $OUT/mix.json > $OUT/mix.txt 2>&1
python3 - <<'PY'
import json
m=json.load(open("gpurun_out/r2k/mix.json"))
for k,v in m.items():
    if "GRBM_GUI_ACTIVE" not in v or not v.get("SQ_INSTS_VALU"): continue
    cyc=v["GRBM_GUI_ACTIVE"]*256
    print("%-28s VALU %.2f SALU %.2f LDS %.3f /cycle/CU; wait_any %.2f wait_inst %.2f of wave cycles; waves %d vgpr %s lds %s grid %s" % (k, v["SQ_INSTS_VALU"]/cyc, v["SQ_INSTS_SALU"]/cyc, v["SQ_INSTS_LDS"]/cyc, v["SQ_WAIT_ANY"]/max(1,v["SQ_WAVE_CYCLES"]), v["SQ_WAIT_INST_ANY"]/max(1,v["SQ_WAVE_CYCLES"]), v["SQ_WAVES"], v["_meta"]["vgpr"], v["_meta"]["lds"], v["_meta"]["grid"]))
    print("      per launch: VALU %.3g SALU %.3g LDS %.3g VMEM_RD %.3g gui_cycles %.3g" % (v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"], v["SQ_INSTS_VMEM_RD"], v["GRBM_GUI_ACTIVE"]))
PY
