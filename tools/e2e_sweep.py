"""Diagnostics: bench.py's e2e leg (slab path) for several host-thread counts."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
import indelope_amd  # noqa: E402
from indelope_amd import synth  # noqa: E402
api = indelope_amd.api()
api.init(0)
b, _ = synth.config("C2")
b = b.with_trim_bounds()
p = api.params(K=27)
for th in (2, 3, 4):
    r = bench.e2e_rates(api, b, p, threads=th, reps=5)
    print(th, r["sustained"], "full", r["full_results"]["sustained"], "arrays", r["pageable_arrays"]["regions_per_s"], r["one_batch_ms"])
