#!/usr/bin/env python3
"""Throughput of the ROI evidence scan (ihp_gen_roi): host buffers in -> regions out, beside
the sequential CPU oracle on one core.  Synthetic run of reads: `coverage`x over `span` bp, 150 bp reads, a share
of them with one indel / clip, hot spots every 5 kb where most reads carry the same deletion."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import indelope_amd  # noqa: E402
from indelope_amd import _abi as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--span", type=int, default=20_000_000)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    n = int(args.span * args.coverage / 150)
    st = np.sort(rng.integers(0, args.span - 200, n)).astype(np.int64)
    # every read: aM kOP bM with a + b = 150 (OP = D at a hot spot, else a random op, else a plain 150M written as 75M 0I 75M)
    a = rng.integers(10, 140, n)
    hot = (st // 5000 + 1) * 5000
    at_hot = (hot > st + 10) & (hot < st + 140) & (rng.random(n) < 0.8)
    a = np.where(at_hot, hot - st, a)
    kind = np.where(at_hot, 2, np.where(rng.random(n) < 0.15, rng.choice([1, 2, 4, 8], n), 1))
    klen = np.where(at_hot, 5, np.where(kind == 1, rng.integers(0, 2, n) * rng.integers(1, 10, n), rng.integers(1, 10, n)))
    cig = np.empty((n, 3), np.uint32)
    cig[:, 0] = a.astype(np.uint32) << 4
    cig[:, 1] = (klen.astype(np.uint32) << 4) | kind.astype(np.uint32)
    cig[:, 2] = (150 - a).astype(np.uint32) << 4
    cons = np.isin(kind, (2, 8))
    en = (st + 150 + np.where(cons, klen, 0)).astype(np.int64)
    off = (np.arange(n + 1, dtype=np.int64) * 3)
    skip = (rng.random(n) < 0.02).astype(np.uint8)
    cigf = np.ascontiguousarray(cig.reshape(-1))
    span = int(en.max() + 10)
    rin = A.RoiIn(n, A.ptr(st, A.i64p), A.ptr(en, A.i64p), A.ptr(skip, A.u8p), A.ptr(off, A.i64p), A.ptr(cigf, A.u32p),
                  0, span, 4, 4, 600)
    api = indelope_amd.api()
    api.init(0)
    ts = []
    for _ in range(args.reps + 1):
        out = A.RoiOut()
        t0 = time.perf_counter()
        rc = api.b.gen_roi(C.byref(rin), C.byref(out))
        ts.append(time.perf_counter() - t0)
        assert rc == 0, rc
        n_roi, n_idx = out.n_roi, out.n_read_idx
        api.b.free_roi(C.byref(out))
    dt = sorted(ts[1:])[len(ts[1:]) // 2]
    in_bytes = st.nbytes + en.nbytes + skip.nbytes + off.nbytes + cigf.nbytes
    res = {"workload": "%d reads (%.0fx, 150 bp) over %d bp" % (n, args.coverage, span), "regions": int(n_roi),
           "region_reads": int(n_idx), "ms": round(dt * 1e3, 2), "reads_per_s": round(n / dt, 1),
           "positions_per_s": round(span / dt, 1), "input_bytes": int(in_bytes)}
    if not args.no_cpu:
        import oracle
        o = oracle.get()
        out = A.RoiOut()
        t0 = time.perf_counter()
        assert o.b.gen_roi(C.byref(rin), C.byref(out)) == 0
        res["cpu_oracle_1thread_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        assert (out.n_roi, out.n_read_idx) == (n_roi, n_idx)
        o.b.free_roi(C.byref(out))
        res["gpu_over_cpu_1thread"] = round(res["cpu_oracle_1thread_ms"] / res["ms"], 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
