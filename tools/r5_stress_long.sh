cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/stress
S=${1:-201}
run() { name=$1; shift; timeout 2400 "$@" > gpurun_out/stress/$name.txt 2>&1; echo "== $name rc=$? : $(tail -1 gpurun_out/stress/$name.txt | cut -c1-200)"; grep -c DIFF gpurun_out/stress/$name.txt; }
run pair_a python tools/ksw_pair_stress.py $S 700
run pair_b python tools/ksw_pair_stress.py $((S+1)) 700
run ksw python tools/ksw_stress.py $S 1000
run regions_params python tools/stress_parity.py 1200 $S params
run regions_lengths python tools/stress_parity.py 600 $((S+2)) lengths
run threads python tools/thread_stress.py 6 60 $S
run fb python tools/fb_stress.py $S 60
run contig python tools/contig_stress.py 15000 $S
run sweep python tools/sweep_stress.py 30 $S
python - <<PY
import sys
sys.path.insert(0, '.')
import bench
print("src_sha16", bench.src_sha16())
PY
