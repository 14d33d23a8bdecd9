"""Diagnostics: one short line from a bench.py JSON line on stdin (value, stage times, oracle check)."""
import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln)
        print(tag, d["config"]["workload"][:3], round(d["value"]), d["ms_per_step"], d["kernel_ms"], d.get("oracle_check", {}).get("identical"))
