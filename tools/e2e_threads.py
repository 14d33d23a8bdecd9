"""Diagnostics: the compact-slab path (ihp_batch_upload_slab2 / run / fetch) driven by T host threads with D batches in flight each:
a thread enqueues upload + run of batch k+D-1 (both return at once) before it waits for and fetches batch k."""
import ctypes as C
import sys
import threading
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import indelope_amd  # noqa: E402
from indelope_amd import _abi as A  # noqa: E402
from indelope_amd import synth  # noqa: E402

api = indelope_amd.api()
api.init(0)
b, _ = synth.config("C2")
b = b.with_trim_bounds()
p = api.params(K=27)
combos = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1x1", "3x1", "2x2", "3x2", "4x2"])]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
MODE = sys.argv[2] if len(sys.argv) > 2 else "fetch"
EAGER = MODE != "lazy"
for kv in sys.argv[4:]:                                  # library switches, e.g. chains=0
    k, v = kv.split("=")
    api.debug_set(**{k: int(v)})
for nth, depth in combos:
    slabs = [[api.make_slab2(b) for _ in range(depth)] for _ in range(nth)]

    def start(sl):
        h = api.batch_upload_slab2(sl, p)
        api.batch_set_fetch(h, no_bases=True, eager=EAGER)
        api.batch_run(h)
        return h

    def finish(h):
        if MODE == "nofetch":
            api.batch_sync(h)
            api.batch_free(h)
            return
        out = A.BatchOut()
        assert api.b.batch_fetch(h, C.byref(out)) == 0
        api.b.free_out(C.byref(out))
        api.batch_free(h)

    gate = threading.Barrier(nth + 1)

    def worker(k):
        for h in [start(slabs[k][j]) for j in range(depth)]:   # warm: pools filled with as many batches in flight as the timed loop keeps
            finish(h)
        gate.wait()
        q = []
        for i in range(N):
            q.append(start(slabs[k][i % depth]))
            if len(q) == depth:
                finish(q.pop(0))
        while q:
            finish(q.pop(0))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(nth)]
    for x in th:
        x.start()
    gate.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print("threads %d x depth %d: %.3f ms per batch (%.2f M regions/s)" % (nth, depth, dt / (N * nth) * 1e3, b.n_regions * N * nth / dt / 1e6))
    for row in slabs:
        for s in row:
            s.free()
