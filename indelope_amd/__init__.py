"""indelope_amd -- MI355X (gfx950) implementation of indelope's per-region hot path.

The package is a thin host layer over one C-ABI shared library
(`indelope_amd/lib/libindelope_hip.so`, built from `indelope_amd/csrc/*.hip` by
`indelope_amd.build`).  There is no CPU fallback: if the library is missing or no
GPU is usable the calls fail loudly.
"""
import ctypes as C
import os

# hardware queues for the streams of the batches in flight (two per batch; the HIP runtime's default of 4 makes launch chains
# that should overlap share a queue every few runs).  Read when the runtime starts, so it is set before anything here touches
# HIP; the caller's own setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _abi
from .host import Api, BatchResult, Contig, IhpError, Match, RegionBatch, unaligned  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libindelope_hip.so")
SYNTH_PATH = os.path.join(HERE, "lib", "libihp_synth.so")

_API = None


def load_library():
    """dlopen the HIP library (works without a GPU; compute calls then return IHP_E_NODEVICE)."""
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built: run `python -m indelope_amd.build` (hipcc, gfx950); "
                          "there is no CPU fallback" % LIB_PATH)
    return C.CDLL(LIB_PATH)


class HipApi(Api):
    def __init__(self):
        self.cdll = load_library()
        super().__init__(_abi.bind(self.cdll, "ihp_", product=True))
        self.cdll.ksw_extz2_sse.restype = None
        self.cdll.ksw_extz2_sse.argtypes = _abi.KSW_ARGTYPES

    def init(self, device=0):
        rc = self.b.init(device)
        if rc != 0:
            raise IhpError(rc, "ihp_init: %s / %s" % (self.b.strerror(rc).decode(), self.b.last_hip_error().decode()))

    def _chk_hip(self, rc, what):
        if rc != 0:
            raise IhpError(rc, "%s: %s / %s" % (what, self.b.strerror(rc).decode(),
                                                  self.b.last_hip_error().decode()))

    # device-resident batches (bench.py, multi-GPU driver)
    def batch_upload(self, batch, params=None):
        p = params if params is not None else self.params()
        cin = batch.as_c()
        h = C.c_void_p()
        self._chk_hip(self.b.batch_upload(C.byref(p), C.byref(cin), C.byref(h)), "batch_upload")
        return h

    def make_slab(self, batch):
        """The batch as one page-locked slab (ihp_slab_layout; 4-bit bases as BAM holds them, trim bounds): what a stager
        fills instead of the separate arrays of ihp_batch_in.  Returns a Slab; free it with slab.free()."""
        return Slab(self, batch)

    def batch_upload_slab(self, slab, params=None):
        p = params if params is not None else self.params()
        h = C.c_void_p()
        self._chk_hip(self.b.batch_upload_slab(C.byref(p), slab.n_regions, slab.n_reads, slab.ptr, C.byref(slab.layout), slab.flags,
                                               C.byref(h)), "batch_upload_slab")
        return h

    def make_slab2(self, batch, bases_2bit=None):
        """The batch as one page-locked COMPACT slab (ihp_slab2_layout: 14 bytes per read, windows and -- when they are all
        A C G T -- read bases 2 bits each, 4 bits otherwise; bases_2bit=False keeps BAM's 4-bit form)."""
        return Slab2(self, batch, bases_2bit)

    def batch_upload_slab2(self, slab, params=None):
        p = params if params is not None else self.params()
        h = C.c_void_p()
        self._chk_hip(self.b.batch_upload_slab2(C.byref(p), slab.n_regions, slab.n_reads, slab.ptr, C.byref(slab.layout), slab.flags,
                                                C.byref(h)), "batch_upload_slab2")
        return h

    def batch_set_fetch(self, h, no_bases=False, eager=False, compact=False):
        fl = (_abi.IHP_FETCH_NO_BASES if no_bases else 0) | (_abi.IHP_FETCH_EAGER if eager else 0) | (_abi.IHP_FETCH_COMPACT if compact else 0)
        self._chk_hip(self.b.batch_set_fetch(h, fl), "batch_set_fetch")

    def batch_run(self, h):
        self._chk_hip(self.b.batch_run(h), "batch_run")

    def batch_sync(self, h):
        self._chk_hip(self.b.batch_sync(h), "batch_sync")

    def batch_fetch(self, h, expand=True):
        """The batch's results.  Under batch_set_fetch(compact=True) they arrive with 4-bit bases and byte supports
        (`compact` attribute of the result); expand=True turns them into the plain arrays through ihp_out_contig."""
        out = _abi.BatchOut()
        self._chk_hip(self.b.batch_fetch(h, C.byref(out)), "batch_fetch")
        try:
            return BatchResult(out, expand_with=self.b.out_contig if expand else None)
        finally:
            self.b.free_out(C.byref(out))

    def batch_stage_ms(self, h):
        ms = (C.c_float * 4)()
        self._chk_hip(self.b.batch_stage_ms(h, ms), "batch_stage_ms")
        return [float(x) for x in ms]

    def batch_pack_dev(self, h):
        """(device pointer, bytes, counts[6]) of the batch's results compacted on the device (no host copy)."""
        import numpy as np
        p, n, counts = C.c_void_p(), C.c_int64(), np.zeros(6, np.int64)
        self._chk_hip(self.b.batch_pack_dev(h, C.byref(p), C.byref(n), _abi.ptr(counts, _abi.i64p)), "batch_pack_dev")
        return p.value, n.value, counts

    def copy_to_host(self, dev_ptr, nbytes):
        """uint8 array with a copy of device memory this library handed out (pack slab, summary records)."""
        import numpy as np
        out = np.zeros(max(1, nbytes), np.uint8)
        self._chk_hip(self.b.copy_to_host(dev_ptr, nbytes, out.ctypes.data_as(C.c_void_p)), "copy_to_host")
        return out[:nbytes]

    def batch_set_timing(self, h, on=True):
        self._chk_hip(self.b.batch_set_timing(h, 1 if on else 0), "batch_set_timing")

    def batch_kernel_ms(self, h):
        """[assemble, ksw2, tally, fallback] execution times (device wall clock) of the last run; needs batch_set_timing."""
        ms = (C.c_float * 4)()
        self._chk_hip(self.b.batch_kernel_ms(h, ms), "batch_kernel_ms")
        return [float(x) for x in ms]

    def batch_kernel_ms_mean(self, h, reset=False):
        """([assemble, ksw2, tally, fallback] mean ms, runs) over the runs synced since timing was switched on / the last reset."""
        ms, n = (C.c_float * 4)(), C.c_int64()
        self._chk_hip(self.b.batch_kernel_ms_mean(h, ms, C.byref(n), 1 if reset else 0), "batch_kernel_ms_mean")
        return [float(x) for x in ms], int(n.value)

    def batch_fallback_ms(self, h):
        ms = C.c_float()
        self._chk_hip(self.b.batch_fallback_ms(h, C.byref(ms)), "batch_fallback_ms")
        return float(ms.value)

    def batch_summary_dev(self, h):
        p, n = C.c_void_p(), C.c_int64()
        self._chk_hip(self.b.batch_summary_dev(h, C.byref(p), C.byref(n)), "batch_summary_dev")
        return p.value, n.value

    def batch_summary_ptr(self, h):
        """The address of the per-region records without waiting for the run (fixed from upload to free)."""
        p, n = C.c_void_p(), C.c_int64()
        self._chk_hip(self.b.batch_summary_ptr(h, C.byref(p), C.byref(n)), "batch_summary_ptr")
        return p.value, n.value

    def batch_summary_host(self, h, n):
        """The per-region summary records of the last run as a numpy array of SUMMARY_DTYPE."""
        import numpy as np
        out = np.zeros(max(n, 1), _abi.SUMMARY_DTYPE)
        self._chk_hip(self.b.batch_summary_host(h, out.ctypes.data_as(C.c_void_p), n), "batch_summary_host")
        return out[:n]

    def batch_profile(self, h):
        import numpy as np
        out = np.zeros(64, np.int64)
        self._chk_hip(self.b.batch_profile(h, _abi.ptr(out, _abi.i64p)), "batch_profile")
        return out

    def batch_release_outputs(self, h):
        """Scratch and result buffers back to the device pool (inputs and summary records stay); the next run takes them again."""
        self._chk_hip(self.b.batch_release_outputs(h), "batch_release_outputs")

    def debug_limits(self, cigar_words=0, events=0, hits=0, ksw_bytes=0):
        """Test hook: cap the device pools of batches uploaded from now on (0 = library sizing)."""
        import numpy as np
        lim = np.array([cigar_words, events, hits, ksw_bytes], np.int64)
        self._chk_hip(self.b.debug_limits(_abi.ptr(lim, _abi.i64p)), "debug_limits")

    def debug_set(self, **kw):
        """Test / diagnostics switches of the library (ihp_debug_set); no arguments = reset all."""
        if not kw:
            self._chk_hip(self.b.debug_set(None, 0), "debug_set")
        for k, v in kw.items():
            self._chk_hip(self.b.debug_set(k.encode(), int(v)), "debug_set(%s)" % k)

    def batch_free(self, h):
        self.b.batch_free(h)


class Slab:
    """A RegionBatch laid out in one ihp_host_alloc'ed slab (see include/indelope_hip.h, ihp_slab_layout)."""

    def __init__(self, api_, batch):
        import numpy as np
        from . import synth
        b = batch if batch.trim_lo is not None else batch.with_trim_bounds()
        self.api, self.n_regions, self.n_reads = api_, b.n_regions, b.n_reads
        self.layout = _abi.SlabLayout()
        api_._chk_hip(api_.b.slab_layout_for(b.n_regions, b.n_reads, len(b.bases), len(b.ref_bases), C.byref(self.layout)), "slab_layout_for")
        self.ptr = api_.b.host_alloc(self.layout.bytes)
        if not self.ptr:
            raise MemoryError("ihp_host_alloc(%d)" % self.layout.bytes)
        mem = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), (self.layout.bytes,))
        L = self.layout

        def put(off, a, dt):
            a = np.ascontiguousarray(a, dt).view(np.uint8).reshape(-1)
            mem[off:off + len(a)] = a
        put(L.region_read_off, b.region_read_off, np.int64); put(L.read_off, b.read_off, np.int64)
        put(L.read_start, b.read_start, np.int64); put(L.read_stop, b.read_stop, np.int64)
        put(L.ref_off, b.ref_off, np.int64); put(L.ref_origin, b.ref_origin, np.int64)
        put(L.trim_lo, b.trim_lo, np.int32); put(L.trim_hi, b.trim_hi, np.int32)
        put(L.mapq, b.mapq, np.uint8); put(L.ref_bases, b.ref_bases, np.uint8)
        self.flags = 0
        if b.read_skip is not None:
            put(L.read_skip, b.read_skip, np.uint8)
            self.flags |= _abi.IHP_SLAB_HAS_SKIP
        lib = synth._lib()
        lib.ihp_synth_pack4.argtypes = [_abi.u8p, _abi.i64p, C.c_int64, C.c_void_p]
        ro = np.ascontiguousarray(b.read_off, np.int64)
        bases = np.ascontiguousarray(b.bases if len(b.bases) else np.zeros(1, np.uint8), np.uint8)
        rc = lib.ihp_synth_pack4(_abi.ptr(bases, _abi.u8p), _abi.ptr(ro, _abi.i64p), b.n_reads, self.ptr + L.bases4)
        if rc != 0:
            self.free()
            raise ValueError("a base that BAM's 4-bit alphabet (=ACMGRSVTWYHKDBN) cannot hold")

    def free(self):
        if self.ptr:
            self.api.b.host_free(self.ptr)
            self.ptr = None


class Slab2:
    """A RegionBatch laid out in one ihp_host_alloc'ed compact slab (include/indelope_hip.h, ihp_slab2_layout): what a stager written
    for this library fills per `roi` (src/indelope.nim:21) instead of the arrays of ihp_batch_in."""

    def __init__(self, api_, batch, bases_2bit=None):
        """bases_2bit: None = the reads 2 bits each when every base is upper-case A C G T (IHP_SLAB2_BASES_2BIT), else 4 bits;
        False = always 4 bits (BAM's own packing)."""
        import numpy as np
        from . import synth
        b = batch if batch.trim_lo is not None else batch.with_trim_bounds()
        self.api, self.n_regions, self.n_reads = api_, b.n_regions, b.n_reads
        ref = np.ascontiguousarray(b.ref_bases, np.uint8)
        code2 = np.full(256, 255, np.uint8)
        code2[[65, 67, 71, 84]] = [0, 1, 2, 3]
        r2 = code2[ref]
        two_bit = not bool((r2 == 255).any())
        self.flags = _abi.IHP_SLAB2_REF_2BIT if two_bit else 0
        rd = np.ascontiguousarray(b.bases if len(b.bases) else np.zeros(1, np.uint8), np.uint8)
        reads_2bit = bases_2bit is not False and len(b.bases) > 0 and not bool((code2[rd] == 255).any())
        if reads_2bit:
            self.flags |= _abi.IHP_SLAB2_BASES_2BIT
        self.layout = _abi.Slab2Layout()
        api_._chk_hip(api_.b.slab2_layout_for(b.n_regions, b.n_reads, len(b.bases), len(ref), self.flags, C.byref(self.layout)), "slab2_layout_for")
        self.ptr = api_.b.host_alloc(self.layout.bytes)
        if not self.ptr:
            raise MemoryError("ihp_host_alloc(%d)" % self.layout.bytes)
        mem = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), (self.layout.bytes,))
        L = self.layout

        def put(off, a, dt):
            a = np.ascontiguousarray(a, dt).view(np.uint8).reshape(-1)
            mem[off:off + len(a)] = a
        ro = np.ascontiguousarray(b.read_off, np.int64)
        rro = np.ascontiguousarray(b.region_read_off, np.int64)
        ln = np.diff(ro)
        region_of = np.repeat(np.arange(b.n_regions), np.diff(rro))
        start_rel = np.asarray(b.read_start, np.int64) - np.asarray(b.ref_origin, np.int64)[region_of]
        span = np.asarray(b.read_stop, np.int64) - np.asarray(b.read_start, np.int64)
        if len(ln) and (ln.max() > 65535 or span.max() > 65535 or span.min() < 0 or abs(start_rel).max() >= 2 ** 31 or
                        np.asarray(b.trim_hi).max() > 65535 or np.asarray(b.trim_lo).min() < 0):
            api_.b.host_free(self.ptr)
            raise ValueError("a read that the compact slab's 16 / 32-bit fields cannot hold")
        put(L.region_read_off, rro, np.int64); put(L.region_base_off, ro[rro], np.int64)
        put(L.ref_off, b.ref_off, np.int64); put(L.ref_origin, b.ref_origin, np.int64)
        put(L.start_rel, start_rel, np.int32); put(L.len, ln, np.uint16); put(L.span, span, np.uint16)
        put(L.trim_lo, b.trim_lo, np.uint16); put(L.trim_hi, b.trim_hi, np.uint16)
        put(L.mapq, b.mapq, np.uint8)
        put(L.rflags, (np.asarray(b.read_skip, np.uint8) & 1) if b.read_skip is not None else np.zeros(b.n_reads, np.uint8), np.uint8)
        # the windows: region r from byte (ref_off[r] >> shift) + r
        fo = np.asarray(b.ref_off, np.int64)
        if two_bit:
            for r in range(b.n_regions):
                w = r2[fo[r]:fo[r + 1]]
                pad = (-len(w)) % 4
                q = np.concatenate([w, np.zeros(pad, np.uint8)]).reshape(-1, 4)
                pk = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
                at = L.ref_packed + (int(fo[r]) >> 2) + r
                mem[at:at + len(pk)] = pk
        else:
            code4 = np.full(256, 255, np.uint8)
            for i, ch in enumerate(b"=ACMGRSVTWYHKDBN"):
                code4[ch] = i
            r4 = code4[ref]
            if (r4 == 255).any():
                api_.b.host_free(self.ptr)
                raise ValueError("a window base that BAM's 4-bit alphabet cannot hold")
            for r in range(b.n_regions):
                w = r4[fo[r]:fo[r + 1]]
                pad = len(w) & 1
                q = np.concatenate([w, np.zeros(pad, np.uint8)]).reshape(-1, 2)
                pk = ((q[:, 0] << 4) | q[:, 1]).astype(np.uint8)
                at = L.ref_packed + (int(fo[r]) >> 1) + r
                mem[at:at + len(pk)] = pk
        lib = synth._lib()
        lib.ihp_synth_pack4.argtypes = [_abi.u8p, _abi.i64p, C.c_int64, C.c_void_p]
        lib.ihp_synth_pack2.argtypes = [_abi.u8p, _abi.i64p, C.c_int64, C.c_void_p]
        bases = rd
        if reads_2bit:
            rc = lib.ihp_synth_pack2(_abi.ptr(bases, _abi.u8p), _abi.ptr(ro, _abi.i64p), b.n_reads, self.ptr + L.bases4)
        else:
            rc = lib.ihp_synth_pack4(_abi.ptr(bases, _abi.u8p), _abi.ptr(ro, _abi.i64p), b.n_reads, self.ptr + L.bases4)
        if rc != 0:
            self.free()
            raise ValueError("a base that BAM's 4-bit alphabet (=ACMGRSVTWYHKDBN) cannot hold")

    def free(self):
        if self.ptr:
            self.api.b.host_free(self.ptr)
            self.ptr = None


def api():
    global _API
    if _API is None:
        _API = HipApi()
    return _API
