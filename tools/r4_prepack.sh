#!/bin/bash
# k_prepack vs k_prepack_fast<1> / <2>: kernel time on a steady 100 000-region run
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for k in ${KS:-0 1 2}; do
  rm -rf gpurun_out/pp_$k
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_$k -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1 --knob prepack_fast=$k > gpurun_out/pp_$k.log 2>&1
  f=$(find gpurun_out/pp_$k -name "*kernel_stats.csv" | head -1)
  echo "prepack_fast=$k: $(grep -i prepack $f | cut -c1-120)"
  rm -rf gpurun_out/pp_$k
done
