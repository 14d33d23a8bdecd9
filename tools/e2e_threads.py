"""Diagnostics: per-phase times of the slab path (ihp_batch_upload_slab / run / fetch) when several host threads drive batches."""
import ctypes as C
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import indelope_amd  # noqa: E402
from indelope_amd import _abi as A  # noqa: E402
from indelope_amd import synth  # noqa: E402

api = indelope_amd.api()
api.init(0)
b, _ = synth.config("C2")
b = b.with_trim_bounds()
p = api.params(K=27)
for nth in (1, 2, 3):
    slabs = [api.make_slab(b) for _ in range(nth)]
    rec = []

    def worker(k):
        for _ in range(8):
            t0 = time.perf_counter()
            h = api.batch_upload_slab(slabs[k], p)
            t1 = time.perf_counter()
            api.batch_run(h)
            t2 = time.perf_counter()
            api.batch_sync(h)
            t3 = time.perf_counter()
            api.batch_set_fetch(h, no_bases=True)
            out = A.BatchOut()
            assert api.b.batch_fetch(h, C.byref(out)) == 0
            t4 = time.perf_counter()
            api.b.free_out(C.byref(out))
            api.batch_free(h)
            t5 = time.perf_counter()
            rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(nth)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    m = np.median(np.array(rec) * 1e3, axis=0)
    print("threads %d: %.3f ms per batch (%.2f M regions/s); median ms upload %.2f launch %.2f wait %.2f fetch %.2f free %.2f"
          % (nth, dt / (8 * nth) * 1e3, b.n_regions * 8 * nth / dt / 1e6, *m))
    for s in slabs:
        s.free()
