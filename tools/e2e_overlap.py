"""Diagnostics: do slab uploads (H2D copies) and resident runs overlap?  Rates of each alone and of both at once."""
import sys
import threading
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import indelope_amd  # noqa: E402
from indelope_amd import synth  # noqa: E402

api = indelope_amd.api()
api.init(0)
b, _ = synth.config("C2")
b = b.with_trim_bounds()
p = api.params(K=27)
slab = api.make_slab(b)
slab2 = api.make_slab(b)
hs = [api.batch_upload_slab(slab2, p) for _ in range(2)]
for h in hs:
    api.batch_run(h)
for h in hs:
    api.batch_sync(h)
N = 20
stop = False


def uploads(res):
    h = api.batch_upload_slab(slab, p); api.batch_sync(h); api.batch_free(h)
    t0 = time.perf_counter()
    n = 0
    while n < N:
        h = api.batch_upload_slab(slab, p)
        api.batch_sync(h)
        api.batch_free(h)
        n += 1
    res["upload_ms"] = (time.perf_counter() - t0) / n * 1e3


def runs(res):
    t0 = time.perf_counter()
    n = 0
    while n < N:
        for h in hs:
            api.batch_run(h)
        for h in hs:
            api.batch_sync(h)
        n += 2
    res["run_ms"] = (time.perf_counter() - t0) / n * 1e3


r = {}
uploads(r); runs(r)
print("alone: upload %.3f ms per batch, run %.3f ms per batch (two chains in flight)" % (r["upload_ms"], r["run_ms"]))
r = {}
ta, tb = threading.Thread(target=uploads, args=(r,)), threading.Thread(target=runs, args=(r,))
ta.start(); tb.start(); ta.join(); tb.join()
print("together: upload %.3f ms per batch, run %.3f ms per batch" % (r["upload_ms"], r["run_ms"]))
