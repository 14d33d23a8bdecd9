"""Diagnostics: value / e2e / other_configs of bench.py JSON lines on stdin."""
import json
import sys
for ln in sys.stdin:
    if ln.startswith("{") and '"metric"' in ln:
        d = json.loads(ln)
        e = d.get("e2e")
        print(round(d["value"]), "e2e", round(e["sustained"]["regions_per_s"]) if e else None,
              {k: (round(v["value"]), v["kernel_ms"]) for k, v in d.get("other_configs", {}).items()})
