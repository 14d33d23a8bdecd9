#!/bin/bash
mkdir -p gpurun_out/diag
IHP_PROFILE=1 timeout 300 python bench.py --no-cpu --no-e2e --no-check --sub-batches 1 --regions 5000 --steps 1 --warmup 0 > gpurun_out/diag/out.txt 2> gpurun_out/diag/err.txt
grep -c DIAG gpurun_out/diag/out.txt
python3 - <<'PY'
import re
rows=[]
for l in open('gpurun_out/diag/out.txt'):
    if l.startswith('DIAG'):
        v=l.split()
        rows.append(dict(r=int(v[2]), total=int(v[4]), takeover=int(v[6]), npre=int(v[12]), nfinal=int(v[14]), t4=int(v[16]), t5=int(v[18]), t6=int(v[20]), t7=int(v[22])))
import statistics as st
tot=sorted(x['total'] for x in rows)
n=len(tot)
print('regions',n,'total cycles: median',tot[n//2],'p90',tot[int(n*.9)],'p99',tot[int(n*.99)],'max',tot[-1])
rows.sort(key=lambda x:-x['total'])
# S.prof are cumulative per wave: show deltas not available; print top rows
for x in rows[:12]: print(x)
import collections
by=collections.defaultdict(list)
for x in rows: by[x['npre']].append(x['total'])
for k in sorted(by): print('npre',k,'n',len(by[k]),'median',sorted(by[k])[len(by[k])//2])
PY
