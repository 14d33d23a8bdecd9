#!/bin/bash
# Instruction-mix / occupancy PMC passes over a short bench run (one counter group per pass).
# usage: tools/pmc_mix.sh [bench args...]   -> gpurun_out/pmc_mix/{g1,g2,g3}, summary in gpurun_out/pmc_mix.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mix
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 2 --no-cpu --no-e2e --no-check --no-other $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/g1 -- python3 bench.py $ARGS > $OUT/g1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/g2 -- python3 bench.py $ARGS > $OUT/g2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $OUT/g3 -- python3 bench.py $ARGS > $OUT/g3.log 2>&1
python3 tools/pmc_sum.py $OUT/g1 $OUT/g2 $OUT/g3 --last ${PMC_LAST:-6} --json $OUT/mix.json > gpurun_out/pmc_mix.txt 2>&1
tail -3 $OUT/g1.log $OUT/g2.log $OUT/g3.log | cut -c1-300
cat gpurun_out/pmc_mix.txt
