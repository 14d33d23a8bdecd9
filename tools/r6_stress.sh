#!/bin/bash
# Round-6 randomised runs on the final sources (one gpurun call): round 5's harnesses + regions of 200-700 reads (the wide combine
# build, the byte-based passes above 640 reads and 64 contigs) + the compact fetch.   tools/r6_stress.sh [seed] [scale]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/stress
S=${1:-61}; X=${2:-1}
run() { name=$1; shift; timeout 2400 "$@" > gpurun_out/stress/$name.txt 2>&1; echo "== $name rc=$? : $(tail -1 gpurun_out/stress/$name.txt | cut -c1-200)"; grep -c DIFF gpurun_out/stress/$name.txt; }
run pair_a python tools/ksw_pair_stress.py $S $((150*X))
run ksw python tools/ksw_stress.py $S $((250*X))
run regions_params python tools/stress_parity.py $((300*X)) $S params
run regions_lengths python tools/stress_parity.py $((200*X)) $((S+2)) lengths
run regions_deep python tools/stress_parity.py $((120*X)) $((S+3)) deep
run regions_deep_params python tools/stress_parity.py $((80*X)) $((S+4)) deep params
run threads python tools/thread_stress.py 6 $((20*X)) $S
run fb python tools/fb_stress.py $S $((40*X))
run contig python tools/contig_stress.py $((4000*X)) $S
run sweep python tools/sweep_stress.py $((8*X)) $S
python - <<PY
import sys
sys.path.insert(0, '.')
import bench
print("src_sha16", bench.src_sha16())
PY
