#!/bin/bash
# kernel time of the ksw2 sweeps on synthetic jobs (rocprofv3 kernel stats per scenario)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for sc in identical subst1 unrelated; do
  rm -rf gpurun_out/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/ksw_pair_time.py ${1:-100000} ${2:-247} $sc > gpurun_out/kt.log 2>&1
  f=$(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
  echo "== $sc"; grep -E "k_ksw" $f | cut -d, -f1-4,6,7
done
rm -rf gpurun_out/kt
