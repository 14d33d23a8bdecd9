#!/bin/bash
# Round-5 randomised runs on the final sources (one gpurun call): the pair sweep (its three hand-written loops), the alignment
# entry point over every band width / flag, whole regions with randomised parameters and lengths, several host threads at once,
# the alignment fallback, the end-to-end sweeps.   tools/r5_stress.sh [seed]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/stress
S=${1:-51}
run() { name=$1; shift; timeout 1500 "$@" > gpurun_out/stress/$name.txt 2>&1; echo "== $name rc=$? : $(tail -1 gpurun_out/stress/$name.txt | cut -c1-200)"; grep -c DIFF gpurun_out/stress/$name.txt; }
run pair_a python tools/ksw_pair_stress.py $S 250
run pair_b python tools/ksw_pair_stress.py $((S+1)) 250
run ksw python tools/ksw_stress.py $S 400
run regions_params python tools/stress_parity.py 400 $S params
run regions_lengths python tools/stress_parity.py 300 $((S+2)) lengths
run threads python tools/thread_stress.py 6 30 $S
run fb python tools/fb_stress.py $S 24
run contig python tools/contig_stress.py 6000 $S
run sweep python tools/sweep_stress.py 12 $S
python - <<PY
import sys
sys.path.insert(0, '.')
import bench
print("src_sha16", bench.src_sha16())
PY
