#!/bin/bash
# Round profile of the default bench workload (C2) on the GPU box:
#   rocprofv3 --kernel-trace --stats  -> gpurun_out/prof/stats  (per-kernel time)
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)   -> gpurun_out/prof/pmc_traffic.json
#   instruction-mix passes (tools/pmc_mix.sh)                   -> gpurun_out/pmc_mix/mix.json
#   bench.py (default flags, with cpu_baseline)                 -> gpurun_out/prof/bench.json
# Copy what should be judged into profiles/ afterwards (see profiles/README.md).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1
python3 tools/pmc_sum.py $OUT/fetch $OUT/write --json $OUT/pmc_raw.json > $OUT/pmc_raw.txt 2>&1
python3 - <<'PY'
import json
raw = json.load(open("gpurun_out/prof/pmc_raw.json"))
out = {"workload": "C2 (10000 regions x 64 x 150bp), bench.py --steps 5 --warmup 2",
       "unit": "bytes per launch",
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = counter(KB) * 1024; "
               "FETCH_SIZE is uncalibrated for narrow accesses on gfx950 (MI355X_MICROARCH.md: reads exactly 1/2 for wide "
               "coalesced streams) so fetch may be under-counted by up to 2x",
       "kernels": {}}
for k, v in raw.items():
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is None or w is None:
        continue
    out["kernels"][k] = {"FETCH_SIZE": int(f * 1024), "WRITE_SIZE": int(w * 1024), "traffic": int((f + w) * 1024),
                         "launches_averaged": v.get("_launches")}
json.dump(out, open("gpurun_out/prof/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
bash tools/pmc_mix.sh > /dev/null 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
