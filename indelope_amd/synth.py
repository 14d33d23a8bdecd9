"""Synthetic candidate regions -- ctypes front end of csrc/synth.cpp."""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from .host import RegionBatch

SEED0 = 0x1DE10BE


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_regions", C.c_int32), ("first_region", C.c_int32),
                ("read_len", C.c_int32), ("n_reads_min", C.c_int32), ("n_reads_max", C.c_int32),
                ("n_events", C.c_int32), ("window_len", C.c_int32), ("event_pos", C.c_int32),
                ("err_rate", C.c_double), ("origin0", C.c_int64), ("origin_step", C.c_int64),
                ("dup_frac", C.c_double)]


# BASELINE.json configs.  K is a path parameter, carried here for convenience.
CONFIGS = {
    "C1": dict(config_id=1, n_regions=1, read_len=150, n_reads=(48, 48), err_rate=0.0, n_events=1, K=27),
    "C2": dict(config_id=2, n_regions=10_000, read_len=150, n_reads=(64, 64), err_rate=1e-3, n_events=1, K=27),
    "C3": dict(config_id=3, n_regions=200_000, read_len=150, n_reads=(16, 256), err_rate=1e-3, n_events=1, K=27),
    "C4": dict(config_id=4, n_regions=5_000_000, read_len=150, n_reads=(64, 64), err_rate=1e-3, n_events=1, K=27),
    "C5": dict(config_id=5, n_regions=10_000, read_len=300, n_reads=(64, 64), err_rate=1e-3, n_events=2, K=31,
               window_len=1400, event_pos=500),
    # Not a BASELINE config: the regions the reference admits above C3's 256 reads -- gen_roi hands over up to 600 reads per roi
    # (src/indelope.nim:483-485, :515) and deep exome coverage is where indelope is run (README.md:5).  n ~ logU[257, 600], 150 bp.
    # Substitution rate 2.5e-4 instead of the other configs' 1e-3: every read with an error inside its overlap opens a contig of
    # its own in the read phase (contig.nim:243-248), and at 1e-3 a pile-up of more than ~450 reads leaves more than the 64
    # contigs the packed path's directory holds (such regions take the byte-based passes; `deep_1e3` measures that mix).
    "deep": dict(config_id=6, n_regions=20_000, read_len=150, n_reads=(257, 600), err_rate=2.5e-4, n_events=1, K=27),
    "deep_1e3": dict(config_id=7, n_regions=20_000, read_len=150, n_reads=(257, 600), err_rate=1e-3, n_events=1, K=27),
}

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libihp_synth.so")
        if not os.path.exists(path):
            from . import build
            build.build()
        _LIB = C.CDLL(path)
        _LIB.ihp_synth_sizes.argtypes = [C.POINTER(SynthCfg), A.i64p, A.i64p, A.i64p]
        _LIB.ihp_synth_fill.argtypes = [C.POINTER(SynthCfg), A.i64p, A.i64p, A.u8p, A.u8p, A.i64p, A.i64p,
                                        A.u8p, A.u8p, A.i64p, A.u8p, A.i64p, A.i32p]
    return _LIB


def generate(n_regions, read_len=150, n_reads=(64, 64), err_rate=0.0, n_events=1, config_id=0,
             first_region=0, window_len=0, event_pos=0, seed=None, origin0=1_000_000, origin_step=10_000,
             dup_frac=0.0, **_):
    """Returns (RegionBatch, truth[n_regions, 4]).  dup_frac > 0 plants that fraction of the events as tandem
    duplications, which send the k-mer tally to the alignment fallback (indelope.nim:312-372)."""
    lib = _lib()
    cfg = SynthCfg(SEED0 ^ config_id if seed is None else seed, n_regions, first_region, read_len,
                   n_reads[0], n_reads[1], n_events, window_len, event_pos, err_rate, origin0, origin_step, dup_frac)
    nr, nb, nf = C.c_int64(), C.c_int64(), C.c_int64()
    lib.ihp_synth_sizes(C.byref(cfg), C.byref(nr), C.byref(nb), C.byref(nf))
    nr, nb, nf = nr.value, nb.value, nf.value
    b = RegionBatch(np.zeros(n_regions + 1, np.int64), np.zeros(nr + 1, np.int64), np.zeros(max(nb, 1), np.uint8),
                    np.zeros(max(nb, 1), np.uint8), np.zeros(max(nr, 1), np.int64), np.zeros(max(nr, 1), np.int64),
                    np.zeros(max(nr, 1), np.uint8), np.zeros(max(nr, 1), np.uint8),
                    np.zeros(n_regions + 1, np.int64), np.zeros(max(nf, 1), np.uint8), np.zeros(max(n_regions, 1), np.int64))
    truth = np.zeros((max(n_regions, 1), 4), np.int32)
    lib.ihp_synth_fill(C.byref(cfg), A.ptr(b.region_read_off, A.i64p), A.ptr(b.read_off, A.i64p),
                       A.ptr(b.bases, A.u8p), A.ptr(b.quals, A.u8p), A.ptr(b.read_start, A.i64p),
                       A.ptr(b.read_stop, A.i64p), A.ptr(b.mapq, A.u8p), A.ptr(b.read_skip, A.u8p),
                       A.ptr(b.ref_off, A.i64p), A.ptr(b.ref_bases, A.u8p), A.ptr(b.ref_origin, A.i64p),
                       A.ptr(truth, A.i32p))
    b.bases, b.quals = b.bases[:nb], b.quals[:nb]
    b.read_start, b.read_stop, b.mapq, b.read_skip = b.read_start[:nr], b.read_stop[:nr], b.mapq[:nr], b.read_skip[:nr]
    b.ref_bases, b.ref_origin = b.ref_bases[:nf], b.ref_origin[:n_regions]
    return b, truth[:n_regions]


def config(name, n_regions=None, first_region=0):
    """A BASELINE.json config (optionally a shard / a reduced region count of it)."""
    kw = dict(CONFIGS[name])
    if n_regions is not None:
        kw["n_regions"] = n_regions
    kw["first_region"] = first_region
    return generate(**kw)
