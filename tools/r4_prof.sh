#!/bin/bash
# kernel stats of a steady 100 000-region run and of the default C2 run (one gpurun call)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_a gpurun_out/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1 "$@" > gpurun_out/prof_a.log 2>&1
f=$(find gpurun_out/prof_a -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/steady_kernel_stats.csv; head -14 gpurun_out/steady_kernel_stats.csv | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python3 bench.py --no-cpu --no-e2e --no-check --no-other --steps 5 --warmup 2 "$@" > gpurun_out/prof_b.log 2>&1
f=$(find gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/c2_kernel_stats.csv; head -14 gpurun_out/c2_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_a gpurun_out/prof_b
