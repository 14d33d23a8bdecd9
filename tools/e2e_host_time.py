import sys, time
sys.path.insert(0, '.')
import indelope_amd
from indelope_amd import synth
api = indelope_amd.api(); api.init(0)
b, _ = synth.config("C2")
b = b.with_trim_bounds()
s2 = api.make_slab2(b)
p = api.params(K=27)
hs = []
for i in range(3):
    h = api.batch_upload_slab2(s2, p); api.batch_run(h); api.batch_sync(h); api.batch_free(h)
api.debug_set(verbose=2)
t0 = time.perf_counter(); h = api.batch_upload_slab2(s2, p); t1 = time.perf_counter(); api.batch_run(h); t2 = time.perf_counter()
api.batch_set_fetch(h, no_bases=True, eager=True)
api.batch_sync(h); t3 = time.perf_counter()
r = api.batch_fetch(h); t4 = time.perf_counter()
api.batch_free(h); t5 = time.perf_counter()
print("upload call %.0f us, run call %.0f us, sync %.0f us, fetch %.0f us, free %.0f us" % tuple(1e6 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)))
