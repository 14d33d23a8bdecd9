#!/usr/bin/env python3
"""Copy the summaries of tools/profile_round.sh (gpurun_out/prof) into profiles/ under the round's prefix.
Files that carry the hash of the library sources they were taken from (`src_sha16` / `_src_sha16`: the PMC passes, the bench
lines, the compiler's resource usage) are REFUSED when that hash is not the one of the sources in the tree: evidence of
another build is not this round's evidence.
usage: tools/collect_profiles.py r04"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
pre = sys.argv[1] if len(sys.argv) > 1 else "r06"
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SHA = bench.src_sha16()


def sha_of(path):
    """The source hash a summary carries, or None when it carries none."""
    try:
        txt = open(path).read()
    except Exception:
        return None
    m = re.search(r'"_?src_sha16"\s*:\s*"([0-9a-f]{16})"', txt) or re.search(r"src_sha16[ =:]+([0-9a-f]{16})", txt)
    return m.group(1) if m else None


names = {"c2_kernel_stats.csv": "c2_kernel_stats.csv", "c2_pmc.json": "c2_pmc.json", "c2_pmc_mix.json": "c2_pmc_mix.json",
         "bench.json": "c2_bench.json", "steady100k_kernel_stats.csv": "steady100k_kernel_stats.csv",
         "steady100k_pmc_mix.json": "steady100k_pmc_mix.json", "steady100k_bench.json": "steady100k_bench.json",
         "c3_kernel_stats.csv": "c3_kernel_stats.csv", "c5_kernel_stats.csv": "c5_kernel_stats.csv",
         "other_workloads.jsonl": "other_workloads.jsonl", "c4_strong_1gpu.json": "c4_strong_1gpu.json",
         "ubench_issue.txt": "ubench_issue.txt", "c2_phase_cycles.json": "c2_phase_cycles.json",
         "c5_phase_cycles.json": "c5_phase_cycles.json", "resource_usage.txt": "resource_usage.txt", "ubench_ksw.txt": "ubench_ksw.txt",
         "ksw_pair_pmc.json": "ksw_pair_pmc.json", "ksw_pair_pmc.txt": "ksw_pair_pmc.txt", "dup10_kernel_stats.csv": "dup10_kernel_stats.csv",
         "dup10_pmc_mix.json": "dup10_pmc_mix.json", "dup10_summary.txt": "dup10_summary.txt", "prepack_compare.txt": "prepack_compare.txt",
         "deep_kernel_stats.csv": "deep_kernel_stats.csv", "deep_bench.json": "deep_bench.json", "ubench_wany.txt": "ubench_wany.txt",
         "e2e_bench.json": "e2e_bench.json", "e2e_host_time.txt": "e2e_host_time.txt", "combine_occupancy.txt": "combine_occupancy.txt"}
refused = 0
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        h = sha_of(p)
        if h is not None and h != SHA:
            print("REFUSED", a, "(taken from sources %s, the tree is %s)" % (h, SHA))
            refused += 1
            continue
        shutil.copy(p, os.path.join(dst, "%s_%s" % (pre, b)))
        print("copied", a, "->", "%s_%s" % (pre, b), "" if h else "(carries no source hash)")
    else:
        print("missing", a)
sys.exit(1 if refused else 0)
