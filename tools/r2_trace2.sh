#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/tr2
for occ in 16; do
IHP_V2_OCC=$occ rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr2/o$occ -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-check --sub-batches 1 --regions 5000 > gpurun_out/tr2/o$occ.log 2>&1
f=$(find gpurun_out/tr2/o$occ -name "*kernel_trace.csv" | head -1)
python3 - "$f" $occ <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: take last 14 kernels
sel=[r for r in rows if 'k_' in r['Kernel_Name']][-16:-7]
t0=int(sel[0]['Start_Timestamp'])
print('occ',sys.argv[2])
for r in sel:
    print('  %-40s start %8.1f us dur %8.1f us grid %s lds %s'%(r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X', r.get('Grid_Size','?')), r.get('LDS_Block_Size','?')))
PY
rm -rf gpurun_out/tr2/o$occ
done
