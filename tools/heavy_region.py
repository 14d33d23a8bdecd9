"""Diagnostics (needs a DIAG build of the combine kernel: status = cycles >> 10): the counters of the heaviest regions."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import indelope_amd
from indelope_amd import synth
api = indelope_amd.api(); api.init(0)
api.debug_set(profile=1)
names = {4: "vote cyc", 5: "exact cyc", 6: "merge cyc", 7: "trim cyc", 32: "bm calls", 33: "cands", 34: "verify passes", 35: "vote scans", 36: "merges",
         37: "filter passes", 38: "q looks", 39: "sup trims", 41: "F room", 42: "F ncorr", 45: "compactions", 46: "vote iters", 47: "vote surv", 49: "T target", 50: "T query", 51: "T flush"}
for cfg, K in (("C5", 31),):
    b, _ = synth.config(cfg)
    b = b.with_trim_bounds()
    p = api.params(K=K)
    h = api.batch_upload(b, p)
    api.batch_run(h); api.batch_sync(h)
    s = api.batch_summary_host(h, b.n_regions)
    cyc = s["status"].astype(np.float64) * 1024
    api.batch_free(h)
    order = np.argsort(-cyc)
    for r in list(order[:3]) + list(order[5000:5002]):
        sub = b.slice(int(r), int(r) + 1)
        h = api.batch_upload(sub, p)
        api.batch_run(h); api.batch_sync(h)
        pc = api.batch_profile(h)
        api.batch_free(h)
        print("region", r, "cycles in batch", int(cyc[r]), "alone", int(pc[2]), {names[k]: int(pc[k]) for k in names if pc[k]})
