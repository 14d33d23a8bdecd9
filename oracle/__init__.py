"""TEST INFRASTRUCTURE ONLY -- loader for the CPU oracle (oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  It binds liboracle.so (the C restatement) to the same host mirror class
the product uses, and optionally oracle/_ref/libksw2_ref*.so (the reference's own
ksw2 C file compiled from /root/reference by oracle/Makefile).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from indelope_amd import _abi as A
from indelope_amd.host import Api, BatchResult

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("IHP_ORACLE_LIB", os.path.join(HERE, "liboracle.so"))   # e.g. liboracle_asan.so (make asan)
REF_DIR = os.path.join(HERE, "_ref")


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(HERE, "..", "include", "indelope_hip.h"))
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(os.path.join(REF_DIR, "libksw2_ref.so"))):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class Oracle(Api):
    def __init__(self):
        build()
        self.cdll = C.CDLL(LIB)
        super().__init__(A.bind(self.cdll, "orc_", product=False))
        d = self.cdll
        d.orc_ksw_extz2.restype, d.orc_ksw_extz2.argtypes = None, A.KSW_ARGTYPES[1:]
        d.orc_ksw_set_variant.argtypes = [C.c_int]
        d.orc_set_ksw_impl.argtypes = [C.c_void_p]
        d.orc_run_regions_mt.restype = C.c_int
        d.orc_run_regions_mt.argtypes = [C.POINTER(A.Params), C.POINTER(A.BatchIn), C.POINTER(A.BatchOut), C.c_int]
        d.orc_read_trim.restype = C.c_int64
        d.orc_read_trim.argtypes = [A.u8p, C.c_int64, C.c_int, A.i64p, A.i64p]
        d.orc_counters.argtypes = [A.i64p]
        d.orc_bench_regions.restype = C.c_int
        d.orc_bench_regions.argtypes = [C.POINTER(A.Params), C.POINTER(A.BatchIn), C.c_int, C.c_int]
        self._ref = {}

    def set_variant(self, v):
        self.cdll.orc_ksw_set_variant(v)

    def ref_lib(self, sse41=False):
        """The compiled reference ksw2 (None if oracle/_ref was never built)."""
        name = "libksw2_ref_sse41.so" if sse41 else "libksw2_ref.so"
        if name not in self._ref:
            path = os.path.join(REF_DIR, name)
            if not os.path.exists(path):
                self._ref[name] = None
            else:
                lib = C.CDLL(path)
                lib.ksw_extz2_sse.restype, lib.ksw_extz2_sse.argtypes = None, A.KSW_ARGTYPES
                self._ref[name] = lib
        return self._ref[name]

    def use_reference_ksw(self, on=True):
        """Route run_regions' alignments through the compiled reference C (cpu baseline)."""
        lib = self.ref_lib() if on else None
        if on and lib is None:
            return False
        self.cdll.orc_set_ksw_impl(C.cast(lib.ksw_extz2_sse, C.c_void_p) if on else None)
        return True

    def ksw_single(self, fn, q_enc, t_enc, m=5, mat=None, gapo=4, gape=1, w=-1, zdrop=-1, flag=0, km=False):
        """One call of a ksw_extz2_sse-shaped function; returns (fields dict, cigar words)."""
        mat = self.matrix() if mat is None else mat
        ez = A.KswExtz()
        q_enc = np.ascontiguousarray(q_enc, np.uint8)
        t_enc = np.ascontiguousarray(t_enc, np.uint8)
        args = [len(q_enc), A.ptr(q_enc, A.u8p), len(t_enc), A.ptr(t_enc, A.u8p), m, A.ptr(mat, A.i8p),
                gapo, gape, w, zdrop, flag, C.byref(ez)]
        if km:
            args = [None] + args
        fn(*args)
        cig = np.array([ez.cigar[i] for i in range(ez.n_cigar)], np.uint32)
        out = dict(max=ez.max, zdropped=ez.zdropped, max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t,
                   mte=ez.mte, mte_q=ez.mte_q, score=ez.score, n_cigar=ez.n_cigar)
        if ez.cigar:
            C.CDLL(None).free(ez.cigar)
        return out, cig

    def ksw(self, q_enc, t_enc, **kw):
        return self.ksw_single(self.cdll.orc_ksw_extz2, q_enc, t_enc, **kw)

    def ksw_ref(self, q_enc, t_enc, sse41=False, **kw):
        return self.ksw_single(self.ref_lib(sse41).ksw_extz2_sse, q_enc, t_enc, km=True, **kw)

    def read_trim(self, quals, min_quality=15):
        q = np.ascontiguousarray(quals, np.uint8)
        lo, hi = C.c_int64(), C.c_int64()
        a = self.cdll.orc_read_trim(A.ptr(q if len(q) else np.zeros(1, np.uint8), A.u8p), len(q), min_quality,
                                    C.byref(lo), C.byref(hi))
        return int(a), int(lo.value), int(hi.value)

    def run_regions_mt(self, batch, params=None, nthreads=1):
        p = params if params is not None else self.params()
        cin = batch.as_c()
        out = A.BatchOut()
        self._chk(self.cdll.orc_run_regions_mt(C.byref(p), C.byref(cin), C.byref(out), nthreads), "run_regions_mt")
        try:
            return BatchResult(out)
        finally:
            self.b.free_out(C.byref(out))

    def bench_regions(self, batch, params=None, nthreads=1, reps=1):
        p = params if params is not None else self.params()
        cin = batch.as_c()
        self._chk(self.cdll.orc_bench_regions(C.byref(p), C.byref(cin), nthreads, reps), "bench_regions")

    def counters(self):
        c = np.zeros(3, np.int64)
        self.cdll.orc_counters(A.ptr(c, A.i64p))
        return dict(compares=int(c[0]), dp_cells=int(c[1]), kmer_steps=int(c[2]))


_ORACLE = None


def get():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
