#!/bin/bash
# per-kernel times of a large single-chain batch (steady state) and of the default C2 step
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r2h; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t100k -- python3 bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1 > $OUT/t100k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 bench.py --no-cpu --no-e2e --no-check --steps 5 --warmup 2 > $OUT/c2.log 2>&1
for d in t100k c2; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; echo "== $d"; cut -d, -f1-4,6,7 $f | head -16; done
