#!/usr/bin/env python3
"""Copy the summaries of tools/profile_round.sh (gpurun_out/prof) into profiles/ under the round's prefix.
usage: tools/collect_profiles.py r03"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
pre = sys.argv[1] if len(sys.argv) > 1 else "r03"
names = {"c2_kernel_stats.csv": "c2_kernel_stats.csv", "c2_pmc.json": "c2_pmc.json", "c2_pmc_mix.json": "c2_pmc_mix.json",
         "bench.json": "c2_bench.json", "steady100k_kernel_stats.csv": "steady100k_kernel_stats.csv",
         "steady100k_pmc_mix.json": "steady100k_pmc_mix.json", "steady100k_bench.json": "steady100k_bench.json",
         "c3_kernel_stats.csv": "c3_kernel_stats.csv", "c5_kernel_stats.csv": "c5_kernel_stats.csv",
         "other_workloads.jsonl": "other_workloads.jsonl", "c4_strong_1gpu.json": "c4_strong_1gpu.json",
         "ubench_issue.txt": "ubench_issue.txt", "c2_phase_cycles.json": "c2_phase_cycles.json",
         "c5_phase_cycles.json": "c5_phase_cycles.json"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (pre, b)))
        print("copied", a, "->", "%s_%s" % (pre, b))
    else:
        print("missing", a)
