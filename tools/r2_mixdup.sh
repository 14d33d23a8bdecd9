#!/bin/bash
bash tools/pmc_mix.sh --no-e2e --no-check --dup-frac 0.1 > /dev/null 2>&1
mkdir -p gpurun_out/mixdup; cp gpurun_out/pmc_mix/mix.json gpurun_out/mixdup/mix.json
rm -rf gpurun_out/pmc_mix/g1 gpurun_out/pmc_mix/g2 gpurun_out/pmc_mix/g3
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/mixdup/mix.json'))
for k,v in d.items():
    if not k.startswith('k_') or 'GRBM_GUI_ACTIVE' not in v: continue
    cu=v['GRBM_GUI_ACTIVE']*32
    if cu < 5e7: continue
    print('%-24s'%k,'cuMcyc %8.1f'%(cu/1e6),'VALUbusy %.2f'%(v.get('SQ_ACTIVE_INST_VALU',0)/cu),'SALUbusy %.2f'%(v.get('SQ_INST_CYCLES_SALU',0)/cu), 'valu %.0fM salu %.0fM lds %.0fM vmem %.0fM'%(v.get('SQ_INSTS_VALU',0)/1e6, v.get('SQ_INSTS_SALU',0)/1e6, v.get('SQ_INSTS_LDS',0)/1e6,(v.get('SQ_INSTS_VMEM_RD',0)+v.get('SQ_INSTS_VMEM_WR',0))/1e6), 'occ %.1f'%(v.get('SQ_WAVE_CYCLES',0)*4/cu), 'waitany %.2f'%(v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1)), v.get('_meta'))
PY
