#!/bin/bash
# round-2 diagnostics: the packed read phase at different VGPR budgets / occupancies
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r2g; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "not c4_shard" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for M in 8 6 4; do
  IHP_ASMR_MINW=$M IHP_PROFILE=1 timeout 300 python3 bench.py --no-cpu --no-e2e --no-check --sub-batches 1 > $OUT/p_m$M.json 2>> $OUT/err
  IHP_ASMR_MINW=$M timeout 300 python3 bench.py --no-cpu --no-e2e --no-check > $OUT/b_m$M.json 2>> $OUT/err
done
IHP_ASM_V1=1 timeout 300 python3 bench.py --no-cpu --no-e2e --no-check > $OUT/b_v1.json 2>> $OUT/err
timeout 300 python3 bench.py --no-cpu --no-e2e --no-check --regions 100000 --steps 3 --warmup 1 --sub-batches 1 > $OUT/t_100k.json 2>> $OUT/err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2g/*.json")):
    try:
        d=json.load(open(f)); p=d.get("profile_cycles"); print(f, d["value"], d["ms_per_step"], d["kernel_ms"], {i:v for i,v in enumerate(p) if v} if p else "")
    except Exception as e: print(f, "ERR", e)
PY
