#!/bin/bash
# the alignment fallback under load: C2 with 10 % tandem duplications; kernel stats, instruction mix, HBM bytes of k_fallback
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
ARGS="--no-cpu --no-e2e --no-check --no-other --dup-frac 0.1 --steps 3 --warmup 1"
O=gpurun_out/fb; rm -rf $O; mkdir -p $O
python3 bench.py $ARGS > $O/plain.log 2>&1; tail -1 $O/plain.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $ARGS > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-150
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/g1 -- python3 bench.py $ARGS > $O/g1.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/g2 -- python3 bench.py $ARGS > $O/g2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/g3 -- python3 bench.py $ARGS > $O/g3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/g4 -- python3 bench.py $ARGS > $O/g4.log 2>&1
python3 tools/pmc_sum.py $O/g1 $O/g2 $O/g3 $O/g4 --last 3 --json $O/mix.json > /dev/null 2>&1
python3 - <<PY
import json
m=json.load(open("$O/mix.json"))
for k,v in m.items():
    if "fallback" in k or "tally" in k:
        print(k[:40], {x: round(y) for x,y in v.items() if isinstance(y,(int,float)) and not x.startswith("_")})
PY
rm -rf $O/kt $O/g1 $O/g2 $O/g3 $O/g4
