#!/usr/bin/env python3
"""PCIe-inclusive rate of the batched path: host buffers in -> host buffers out through
ihp_batch_upload + ihp_batch_run + ihp_batch_fetch, per stage (total = those three C calls; the Python host's
numpy copy of the results and the batch free are listed beside it).  Never bench.py's `value`."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctypes as C  # noqa: E402

import indelope_amd  # noqa: E402
from indelope_amd import _abi as A  # noqa: E402
from indelope_amd.host import BatchResult  # noqa: E402
from indelope_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--regions", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--pinned", action="store_true", help="stage the input arrays in ihp_host_alloc memory")
    ap.add_argument("--quals", action="store_true", help="upload base qualities (trim on the device) instead of trim bounds")
    ap.add_argument("--threads", type=int, default=3,
                    help="host threads for the sustained leg: each runs upload -> run -> fetch on batches of its own, so one "
                         "batch's copies overlap another's kernels (every batch has its own stream)")
    args = ap.parse_args()
    api = indelope_amd.api()
    api.init(0)
    cfg = dict(synth.CONFIGS[args.config])
    if args.regions:
        cfg["n_regions"] = args.regions
    batch, _ = synth.generate(**cfg)
    p = api.params(K=cfg["K"])
    if not args.quals:
        batch = batch.with_trim_bounds()
    if args.pinned:
        import numpy as np
        for f in ("region_read_off", "read_off", "bases", "quals", "read_start", "read_stop", "mapq", "read_skip",
                  "ref_off", "ref_bases", "ref_origin", "trim_lo", "trim_hi"):
            if getattr(batch, f) is None:
                continue
            a = np.ascontiguousarray(getattr(batch, f))
            ptr = api.b.host_alloc(max(1, a.nbytes))
            assert ptr
            v = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (max(1, a.nbytes),))[:a.nbytes].view(a.dtype)
            v[...] = a
            setattr(batch, f, v)
    t = {"upload": [], "run": [], "fetch": [], "py_copy": [], "free": [], "total": []}
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        h = api.batch_upload(batch, p)
        t1 = time.perf_counter()
        api.batch_run(h)
        api.batch_sync(h)
        t2 = time.perf_counter()
        out = A.BatchOut()                                   # the C call alone, then the numpy copy the Python host makes
        rc = api.b.batch_fetch(h, C.byref(out))
        assert rc == 0, rc
        t3 = time.perf_counter()
        res = BatchResult(out)
        api.b.free_out(C.byref(out))
        t4 = time.perf_counter()
        api.batch_free(h)
        t5 = time.perf_counter()
        for k, v in zip(("upload", "run", "fetch", "py_copy", "free", "total"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t3 - t0)):
            t[k].append(v * 1e3)
    med = {k: sorted(v[1:])[len(v[1:]) // 2] for k, v in t.items()}
    sustained = None
    if args.threads > 1:
        import threading
        n_each = max(4, args.reps * 2)

        def worker():
            for _ in range(n_each):
                out = A.BatchOut()
                cin = batch.as_c()
                assert api.b.run_regions(C.byref(p), C.byref(cin), C.byref(out)) == 0
                api.b.free_out(C.byref(out))
        worker()                                             # warm the pools
        th = [threading.Thread(target=worker) for _ in range(args.threads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        sustained = {"threads": args.threads, "batches": n_each * args.threads,
                     "ms_per_batch": round(dt / (n_each * args.threads) * 1e3, 3),
                     "regions_per_s": round(batch.n_regions * n_each * args.threads / dt, 1)}
    print(json.dumps({"workload": args.config, "read_trim": "device (qualities uploaded)" if args.quals else "stager (trim bounds)", "inputs": "pinned (ihp_host_alloc)" if args.pinned else "pageable", "regions": batch.n_regions, "ms": {k: round(v, 3) for k, v in med.items()},
                      "regions_per_s_pcie_inclusive": round(batch.n_regions / (med["total"] * 1e-3), 1),
                      "sustained_ihp_run_regions": sustained,
                      "contigs": int(res.n_contigs), "events": int(res.n_events)}))


if __name__ == "__main__":
    main()
