#!/bin/bash
# Round profile on the GPU box (one gpurun call).  Everything lands in gpurun_out/prof; copy what should be judged into
# profiles/ afterwards (tools/collect_profiles.py does, see profiles/README.md).
#   default C2 bench under rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE / WRITE_SIZE (separate passes), the
#   instruction-mix passes (tools/pmc_mix.sh), the plain bench line (with cpu_baseline, e2e, oracle check);
#   steady state (100 000 regions, one chain): kernel stats + bench line + instruction mix;
#   C3 / C5 kernel stats and bench lines, C2 variants (--quals, --dup-frac 0.1, --sub-batches 1), a C4 strong-scaling
#   run on this one GPU (1.25 M regions in 8 resident chunks), the issue-rate microbenchmark.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu --no-e2e --no-check --no-other"
stats() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py "$@" > $OUT/$name.log 2>&1; f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv; }
stats c2 $ARGS
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1
python3 tools/pmc_sum.py $OUT/fetch $OUT/write --last 10 --json $OUT/pmc_raw.json > $OUT/pmc_raw.txt 2>&1
bash tools/pmc_mix.sh > /dev/null 2>&1
cp gpurun_out/pmc_mix/mix.json $OUT/c2_pmc_mix.json
STEADY="--no-cpu --no-e2e --no-check --no-other --regions 100000 --steps 3 --warmup 1 --in-flight 1 --sub-batches 1"
stats steady100k $STEADY
PMC_LAST=3 bash tools/pmc_mix.sh --regions 100000 --in-flight 1 --sub-batches 1 > /dev/null 2>&1
cp gpurun_out/pmc_mix/mix.json $OUT/steady100k_pmc_mix.json
python3 bench.py $STEADY > $OUT/steady100k_bench.json 2>> $OUT/err
stats c3 --config C3 --steps 3 --warmup 1 --no-cpu --no-e2e --no-check
stats c5 --config C5 --steps 5 --warmup 2 --no-cpu --no-e2e --no-check
# round 6: the regions of 257-600 reads (wide combine build): the byte-based k_assemble passes must stay empty
stats deep --config deep --regions 10000 --steps 3 --warmup 2 --in-flight 1 --sub-batches 2 --no-cpu --no-e2e --no-check
python3 bench.py --config deep --regions 10000 --steps 3 --warmup 2 --in-flight 1 --sub-batches 2 --no-cpu --no-e2e > $OUT/deep_bench.json 2>> $OUT/err
timeout 120 tools/ubench_wany.bin > $OUT/ubench_wany.txt 2>&1
: > $OUT/other_workloads.jsonl
python3 bench.py --config C3 --steps 3 --warmup 1 --no-cpu --no-e2e >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --config C5 --no-cpu --no-e2e >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --no-cpu --no-e2e --no-check --no-other --quals >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --no-cpu --no-e2e --no-check --no-other --dup-frac 0.1 >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --no-cpu --no-e2e --no-check --no-other --in-flight 1 --sub-batches 1 >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --no-cpu --no-e2e --no-check --no-other --in-flight 1 --lockstep >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --knob asm_v1=1 --no-cpu --no-e2e --no-check --no-other >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --config C3 --knob no_rich=1 --steps 3 --warmup 1 --no-cpu --no-e2e --no-check >> $OUT/other_workloads.jsonl 2>> $OUT/err
python3 bench.py --scaling strong --config C4 --regions 1250000 --steps 2 --warmup 1 --no-cpu --no-e2e > $OUT/c4_strong_1gpu.json 2>> $OUT/err
timeout 120 tools/ubench_issue.bin > $OUT/ubench_issue.txt 2>&1
timeout 120 tools/ubench_ksw.bin > $OUT/ubench_ksw.txt 2>&1
# what the compiler gave every kernel (registers, spills, scratch, LDS) -- from THESE sources, with their hash in the first line
python3 - <<'PY' > $OUT/resource_usage.txt 2> $OUT/resource_usage.err
import os, re, subprocess, sys
sys.path.insert(0, ".")
import bench
inc, src = "include", "indelope_amd/csrc"
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", inc, "-I", src, "-Wno-unused-function",
                    "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/ihp_ru.so", src + "/indelope_hip.hip"], capture_output=True, text=True)
print("src_sha16 %s  (hipcc -O3 -Rpass-analysis=kernel-resource-usage over indelope_amd/csrc/indelope_hip.hip)" % bench.src_sha16())
cur = None
for ln in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*) \[-Rpass-analysis", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        p = subprocess.run(["c++filt", cur], capture_output=True, text=True) if os.path.exists("/usr/bin/c++filt") else None
        print("\n" + (p.stdout.strip() if p and p.stdout.strip() else cur))
    elif cur:
        print("    " + t)
PY
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --no-cpu --no-e2e --no-other --no-check --profile --in-flight 1 --sub-batches 1 > $OUT/c2_phase_cycles.json 2>> $OUT/err
python3 bench.py --config C5 --no-cpu --no-e2e --no-other --no-check --profile --in-flight 1 --sub-batches 1 > $OUT/c5_phase_cycles.json 2>> $OUT/err
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
sha = bench.src_sha16()
for f in ("gpurun_out/prof/c2_pmc_mix.json", "gpurun_out/prof/steady100k_pmc_mix.json"):
    try:
        m = json.load(open(f)); m["_src_sha16"] = sha; json.dump(m, open(f, "w"), indent=1)
    except Exception as e:
        print("mix", f, e)
raw = json.load(open("gpurun_out/prof/pmc_raw.json"))
out = {"workload": "C2 (two resident batches of 10000 regions x 64 x 150bp taking turns), bench.py --steps 5 --warmup 2: one launch of a kernel = one batch = 10000 regions",
       "unit": "bytes per launch", "src_sha16": sha,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = counter(KB) * 1024; traffic = "
               "2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read "
               "(MI355X_MICROARCH.md, HBM section: double it); narrower accesses and WRITE_SIZE are uncalibrated, so the corrected "
               "figure is an upper bound for the byte-wise kernels; `traffic_raw` = FETCH_SIZE + WRITE_SIZE as counted",
       "kernels": {}}
for k, v in raw.items():
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    if f is None or w is None:
        continue
    out["kernels"][k] = {"FETCH_SIZE": int(f * 1024), "WRITE_SIZE": int(w * 1024), "traffic": int((2 * f + w) * 1024), "traffic_raw": int((f + w) * 1024),
                         "launches_averaged": v.get("_launches")}
json.dump(out, open("gpurun_out/prof/c2_pmc.json", "w"), indent=1)
PY
# round 5: the PCIe-inclusive leg on the compact slab (host buffers in -> host buffers out), the host's share of an upload
python3 bench.py --no-cpu --no-other --no-check > $OUT/e2e_bench.json 2>> $OUT/err
python3 tools/e2e_host_time.py > $OUT/e2e_host_time.txt 2>&1
# round 4: the alignment fallback under load (C2 with 10 % duplications), k_prepack vs k_prepack_fast, the pair sweep's instruction mix
bash tools/r4_fb.sh > $OUT/dup10_summary.txt 2>&1
cp gpurun_out/fb/kernel_stats.csv $OUT/dup10_kernel_stats.csv; cp gpurun_out/fb/mix.json $OUT/dup10_pmc_mix.json
KS="0 2" bash tools/r4_prepack.sh > $OUT/prepack_compare.txt 2>&1
bash tools/r4_kpmc.sh > $OUT/ksw_pair_pmc.txt 2>&1
cp gpurun_out/kp/mix.json $OUT/ksw_pair_pmc.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
sha = bench.src_sha16()
for f in ("gpurun_out/prof/dup10_pmc_mix.json", "gpurun_out/prof/ksw_pair_pmc.json"):
    try:
        m = json.load(open(f)); m["_src_sha16"] = sha; json.dump(m, open(f, "w"), indent=1)
    except Exception as e:
        print("stamp", f, e)
for f in ("gpurun_out/prof/dup10_summary.txt", "gpurun_out/prof/prepack_compare.txt", "gpurun_out/prof/ksw_pair_pmc.txt"):
    try:
        t = open(f).read(); open(f, "w").write("src_sha16 %s\n" % sha + t)
    except Exception as e:
        print("stamp", f, e)
PY
# round 5: k_asm_combine3 against the regions a CU holds (whole LDS granules: 12, 14, 16, 18, 21 per CU)
bash tools/r5_occ.sh "comb_occ=10" "comb_occ=12" "comb_occ=14" "comb_occ=16" "comb_occ=18" "" "comb_minw=6" "comb_minw=6 v2_arena=2768 asm_waves=24" > $OUT/combine_occupancy.txt 2>&1
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
f = "gpurun_out/prof/combine_occupancy.txt"
t = open(f).read(); open(f, "w").write("src_sha16 %s  (tools/r5_occ.sh: steady 100 000-region launches; [knobs]: regions/s, ms per step; the tier line; k_asm_combine3's average launch)\n" % bench.src_sha16() + t)
PY
# the raw rocprofv3 output directories are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $OUT/c2 $OUT/steady100k $OUT/c3 $OUT/c5 $OUT/fetch $OUT/write gpurun_out/pmc_mix/g1 gpurun_out/pmc_mix/g2 gpurun_out/pmc_mix/g3
du -sh gpurun_out
cat $OUT/bench.json | head -c 1500
head -14 $OUT/c2_kernel_stats.csv | cut -c1-150
