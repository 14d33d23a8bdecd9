"""ctypes mirror of include/indelope_hip.h.

Everything here is a plain description of the C ABI: structure layouts and
function prototypes.  `bind(cdll, prefix)` attaches prototypes to a loaded
library; the product library exports the `ihp_` names, the test oracle exports
the same signatures under `orc_` (see oracle/oracle.h), so parity tests drive
both through identical code.
"""
import ctypes as C

import numpy as np

IHP_OK = 0
IHP_E_NODEVICE, IHP_E_HIP, IHP_E_ARG, IHP_E_NOMEM, IHP_E_CAPACITY, IHP_E_UNSUPPORTED = -1, -2, -3, -4, -5, -6
IHP_UNALIGNED = -(2 ** 63)            # contig.nim:27
IHP_ALLOW_DEFAULT, IHP_ALLOW_SUPPORT = 0, 1
KSW_NEG_INF = -0x40000000
KSW_EZ_SCORE_ONLY, KSW_EZ_RIGHT, KSW_EZ_GENERIC_SC = 0x01, 0x02, 0x04
KSW_EZ_APPROX_MAX, KSW_EZ_APPROX_DROP, KSW_EZ_EXTZ_ONLY, KSW_EZ_REV_CIGAR = 0x08, 0x10, 0x40, 0x80
IHP_GT_HOM_REF, IHP_GT_HET, IHP_GT_HOM_ALT, IHP_GT_UNKNOWN = 0, 1, 2, 3
IHP_EV_TALLIED, IHP_EV_SHORT, IHP_EV_SAME_KMER, IHP_EV_LOW_CPLX = 0, 1, 2, 3
IHP_EV_BUG_SAME, IHP_EV_OOB, IHP_EV_NON_ACGT = 4, 5, 6
IHP_ALN_DONE, IHP_ALN_REF_CLAMPED = 1, 2

i8p, u8p = C.POINTER(C.c_int8), C.POINTER(C.c_uint8)
i32p, u32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)


class KswExtz(C.Structure):           # ksw2.h:22-30
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int),
                ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int),
                ("cigar", u32p)]

    @property
    def max(self):
        return self.max_zd & 0x7FFFFFFF

    @property
    def zdropped(self):
        return self.max_zd >> 31


class Ez(C.Structure):                # ihp_ez
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t",
                                          "mte", "mte_q", "score", "n_cigar")]


EZ_DTYPE = np.dtype([(n, "<i4") for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t",
                                           "mte", "mte_q", "score", "n_cigar")])


class Contig(C.Structure):            # ihp_contig
    _fields_ = [("sequence", u8p), ("support", u32p), ("len", C.c_int64), ("cap", C.c_int64),
                ("nreads", C.c_int64), ("start", C.c_int64)]


class Correction(C.Structure):
    _fields_ = [("qoff", C.c_int64), ("toff", C.c_int64), ("qbest", C.c_int32), ("_pad", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("matches", C.c_int64), ("offset", C.c_int64), ("mismatches", C.c_int64),
                ("n_corrections", C.c_int64), ("contig_i", C.c_int64),
                ("corrections", C.POINTER(Correction)), ("corr_cap", C.c_int64)]


class Genotype(C.Structure):
    _fields_ = [("gt", C.c_int32), ("_pad", C.c_int32), ("gl", C.c_double * 3)]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("min_overlap_pct", C.c_double),
                ("min_mapq_assemble", C.c_int32), ("min_mapq_stop", C.c_int32),
                ("min_mapq_tally", C.c_int32), ("trim_min_qual", C.c_int32),
                ("combine_min_support", C.c_int32), ("combine_min_overlap", C.c_int32),
                ("max_mismatch", C.c_int32), ("max_pre_contigs", C.c_int32),
                ("min_ctg_len", C.c_int32), ("min_reads", C.c_int32), ("min_event_len", C.c_int32),
                ("K", C.c_int32), ("max_events", C.c_int32), ("ref_pad", C.c_int32),
                ("match", C.c_int8), ("mismatch", C.c_int8), ("gap_open", C.c_int8), ("gap_ext", C.c_int8),
                ("bw", C.c_int32), ("zdrop", C.c_int32), ("ksw_flag", C.c_int32),
                ("error", C.c_double),
                ("fallback", C.c_int32),
                ("fb_match", C.c_int8), ("fb_mismatch", C.c_int8), ("fb_gap_open", C.c_int8), ("fb_gap_ext", C.c_int8),
                ("fb_bw", C.c_int32), ("fb_zdrop", C.c_int32), ("fb_flag", C.c_int32)]


class BatchIn(C.Structure):
    _fields_ = [("n_regions", C.c_int32), ("n_reads", C.c_int64),
                ("region_read_off", i64p), ("read_off", i64p), ("bases", u8p), ("quals", u8p),
                ("read_start", i64p), ("read_stop", i64p), ("mapq", u8p), ("read_skip", u8p),
                ("ref_off", i64p), ("ref_bases", u8p), ("ref_origin", i64p),
                ("trim_lo", i32p), ("trim_hi", i32p)]


class SlabLayout(C.Structure):        # ihp_slab_layout
    _fields_ = [(n, C.c_int64) for n in ("region_read_off", "read_off", "read_start", "read_stop", "ref_off", "ref_origin",
                                          "trim_lo", "trim_hi", "mapq", "read_skip", "ref_bases", "bases4", "bytes")]


class Slab2Layout(C.Structure):       # ihp_slab2_layout
    _fields_ = [(n, C.c_int64) for n in ("region_read_off", "region_base_off", "ref_off", "ref_origin", "start_rel", "len", "span",
                                          "trim_lo", "trim_hi", "mapq", "rflags", "ref_packed", "bases4", "bytes")]


IHP_SLAB_HAS_SKIP, IHP_FETCH_NO_BASES, IHP_FETCH_EAGER, IHP_FETCH_COMPACT = 1, 1, 2, 4
IHP_SLAB2_REF_2BIT, IHP_SLAB2_BASES_2BIT = 2, 4


class Event(C.Structure):
    _fields_ = [("tstart", C.c_int64), ("tstop", C.c_int64), ("qstart", C.c_int64), ("qstop", C.c_int64),
                ("len", C.c_uint32), ("type", C.c_uint8), ("status", C.c_uint8),
                ("fallback_needed", C.c_uint8), ("aligned", C.c_uint8),
                ("cf_offset", C.c_int32), ("ref_support", C.c_int32), ("alt_support", C.c_int32),
                ("both_found", C.c_int32), ("ref_kmer", C.c_char * 32), ("alt_kmer", C.c_char * 32),
                ("gt", C.c_int32), ("kmer_ref_support", C.c_int32), ("kmer_alt_support", C.c_int32),
                ("kmer_both_found", C.c_int32), ("gl", C.c_double * 3), ("qual", C.c_double)]


EVENT_DTYPE = np.dtype({
    "names": ["tstart", "tstop", "qstart", "qstop", "len", "type", "status", "fallback_needed", "aligned",
              "cf_offset", "ref_support", "alt_support", "both_found", "ref_kmer", "alt_kmer",
              "gt", "kmer_ref_support", "kmer_alt_support", "kmer_both_found", "gl", "qual"],
    "formats": ["<i8", "<i8", "<i8", "<i8", "<u4", "u1", "u1", "u1", "u1",
                "<i4", "<i4", "<i4", "<i4", "S32", "S32", "<i4", "<i4", "<i4", "<i4", ("<f8", 3), "<f8"],
    "offsets": [Event.tstart.offset, Event.tstop.offset, Event.qstart.offset, Event.qstop.offset,
                Event.len.offset, Event.type.offset, Event.status.offset, Event.fallback_needed.offset,
                Event.aligned.offset, Event.cf_offset.offset, Event.ref_support.offset, Event.alt_support.offset,
                Event.both_found.offset, Event.ref_kmer.offset, Event.alt_kmer.offset,
                Event.gt.offset, Event.kmer_ref_support.offset, Event.kmer_alt_support.offset,
                Event.kmer_both_found.offset, Event.gl.offset, Event.qual.offset],
    "itemsize": C.sizeof(Event)})


class BatchOut(C.Structure):
    _fields_ = [("n_regions", C.c_int32), ("n_contigs", C.c_int64), ("n_events", C.c_int64),
                ("n_cigar_words", C.c_int64), ("n_bases", C.c_int64), ("n_hits", C.c_int64),
                ("status", i32p), ("n_contigs_pre", i32p), ("contig_off", i64p),
                ("ctg_start", i64p), ("ctg_nreads", i64p), ("ctg_seq_off", i64p),
                ("ctg_seq", u8p), ("ctg_support", u32p),
                ("aln_flags", i32p), ("aln_ref_start", i64p), ("aln_ref_len", i32p),
                ("aln_ez", C.POINTER(Ez)), ("cigar_off", i64p), ("cigar", u32p),
                ("event_off", i64p), ("events", C.POINTER(Event)),
                ("hit_off", i64p), ("ref_hit", i32p), ("alt_hit", i32p),
                # IHP_FETCH_COMPACT: bases 4 bits each, supports a byte each + escapes (ctg_seq / ctg_support are then NULL)
                ("ctg_seq4", u8p), ("ctg_sup8", u8p), ("n_sup_escapes", C.c_int64), ("sup_escape_idx", i64p), ("sup_escape_val", u32p)]


class Variant(C.Structure):           # ihp_variant
    _fields_ = [("region", C.c_int32), ("contig", C.c_int32), ("event", C.c_int64), ("filter", C.c_int32), ("gt", C.c_int32),
                ("start", C.c_int64), ("qual", C.c_double), ("gq", C.c_double), ("gl", C.c_double * 3),
                ("ake", C.c_double), ("rke", C.c_double), ("ad", C.c_int32 * 2),
                ("dp", C.c_int32), ("bs", C.c_int32), ("mf", C.c_int32), ("cf", C.c_int32), ("nc", C.c_int32),
                ("amq", C.c_int32), ("rmq", C.c_int32),
                ("lo", C.c_uint8), ("al", C.c_uint8), ("event_type", C.c_uint8), ("_pad", C.c_uint8),
                ("ref_len", C.c_int32), ("alt_len", C.c_int32), ("cc_len", C.c_int32),
                ("ref_off", C.c_int64), ("alt_off", C.c_int64), ("cc_off", C.c_int64),
                ("ref_kmer", C.c_char * 32), ("alt_kmer", C.c_char * 32)]


class Variants(C.Structure):          # ihp_variants
    _fields_ = [("n", C.c_int64), ("v", C.POINTER(Variant)), ("n_chars", C.c_int64), ("chars", C.POINTER(C.c_char))]


IHP_VF_EMITTED, IHP_VF_LOW_ALT, IHP_VF_LOW_FRAC, IHP_VF_HOM_REF, IHP_VF_BOTH_AT_EDGE = 0, 2, 3, 4, 5
IHP_VF_SMALL_FLANK, IHP_VF_KMER_AT_END, IHP_VF_HOMOPOLYMER, IHP_VF_DUPLICATE, IHP_VF_OOB = 6, 7, 8, 9, 10


class RoiIn(C.Structure):             # ihp_roi_in
    _fields_ = [("n_reads", C.c_int64), ("read_start", i64p), ("read_stop", i64p), ("read_skip", u8p),
                ("cigar_off", i64p), ("cigar", u32p), ("origin", C.c_int64), ("span", C.c_int64),
                ("min_event_support", C.c_int32), ("min_read_coverage", C.c_int32), ("max_read_coverage", C.c_int32)]


class RoiOut(C.Structure):            # ihp_roi_out
    _fields_ = [("n_roi", C.c_int64), ("n_read_idx", C.c_int64), ("roi_start", i64p), ("roi_stop", i64p),
                ("read_off", i64p), ("reads", i64p)]


class RegionSummary(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("status", "n_contigs_pre", "n_contigs", "n_aligned",
                                          "n_events", "n_tallied", "ref_support", "alt_support")]


SUMMARY_DTYPE = np.dtype([(n, "<i4") for n in ("status", "n_contigs_pre", "n_contigs", "n_aligned",
                                                "n_events", "n_tallied", "ref_support", "alt_support")])

# name -> (restype, argtypes); names without prefix.
_COMMON = {
    "ksw_extz2_batch": (C.c_int, [C.c_int32, u8p, i64p, u8p, i64p, C.c_int8, i8p, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int, C.c_int, C.POINTER(Ez), u32p, C.c_int64, i64p]),
    "encode": (None, [u8p, C.c_int64, u8p]),
    "matrix": (None, [C.c_int8, C.c_int8, i8p]),
    "slide_align": (C.c_int, [C.POINTER(Contig), C.POINTER(Contig), C.c_int64, C.c_int64, C.c_int,
                              C.POINTER(Match)]),
    "contig_insert": (C.c_int, [C.POINTER(Contig), C.POINTER(Contig), C.POINTER(Match)]),
    "contig_trim": (C.c_int, [C.POINTER(Contig), C.c_int64]),
    "kmer_tally": (C.c_int, [C.c_int32, u8p, i64p, u8p, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p,
                             i32p]),
    "genotype": (C.c_int, [C.c_int64, C.c_int64, C.c_double, C.POINTER(Genotype)]),
    "genotype_qual": (C.c_double, [C.POINTER(Genotype)]),
    "params_default": (None, [C.POINTER(Params)]),
    "run_regions": (C.c_int, [C.POINTER(Params), C.POINTER(BatchIn), C.POINTER(BatchOut)]),
    "free_out": (None, [C.POINTER(BatchOut)]),
    "gen_roi": (C.c_int, [C.POINTER(RoiIn), C.POINTER(RoiOut)]),
    "free_roi": (None, [C.POINTER(RoiOut)]),
    "call_variants": (C.c_int, [C.POINTER(Params), C.POINTER(BatchIn), C.POINTER(BatchOut), C.POINTER(Variants)]),
    "free_variants": (None, [C.POINTER(Variants)]),
    "format_variant": (C.c_int64, [C.POINTER(Variant), C.POINTER(C.c_char), C.c_char_p, C.c_char_p, C.c_int64]),
}

_PRODUCT_ONLY = {
    "strerror": (C.c_char_p, [C.c_int]),
    "last_hip_error": (C.c_char_p, []),
    "version": (C.c_char_p, []),
    "init": (C.c_int, [C.c_int]),
    "device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), i64p]),
    "shutdown": (None, []),
    "host_alloc": (C.c_void_p, [C.c_size_t]),
    "host_free": (None, [C.c_void_p]),
    "copy_to_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "batch_upload": (C.c_int, [C.POINTER(Params), C.POINTER(BatchIn), C.POINTER(C.c_void_p)]),
    "slab_layout_for": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.POINTER(SlabLayout)]),
    "batch_upload_slab": (C.c_int, [C.POINTER(Params), C.c_int32, C.c_int64, C.c_void_p, C.POINTER(SlabLayout), C.c_int32,
                                    C.POINTER(C.c_void_p)]),
    "slab2_layout_for": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(Slab2Layout)]),
    "batch_upload_slab2": (C.c_int, [C.POINTER(Params), C.c_int32, C.c_int64, C.c_void_p, C.POINTER(Slab2Layout), C.c_int32,
                                     C.POINTER(C.c_void_p)]),
    "batch_set_fetch": (C.c_int, [C.c_void_p, C.c_int32]),
    "batch_run": (C.c_int, [C.c_void_p]),
    "batch_sync": (C.c_int, [C.c_void_p]),
    "batch_fetch": (C.c_int, [C.c_void_p, C.POINTER(BatchOut)]),
    "out_contig": (C.c_int, [C.POINTER(BatchOut), C.c_int64, u8p, u32p]),
    "batch_free": (None, [C.c_void_p]),
    "batch_release_outputs": (C.c_int, [C.c_void_p]),
    "batch_pack_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i64p, i64p]),
    "unpack_slab": (C.c_int, [C.c_void_p, C.c_int64, i64p, C.c_double, C.POINTER(BatchOut)]),
    "pack_out": (C.c_int, [C.POINTER(BatchOut), C.c_void_p, C.c_int64, i64p, i64p]),
    "batch_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "batch_fallback_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "batch_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "batch_kernel_ms_mean": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), i64p, C.c_int]),
    "batch_summary_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i64p]),
    "batch_summary_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i64p]),
    "batch_profile": (C.c_int, [C.c_void_p, i64p]),
    "batch_profile_n": (C.c_int, [C.c_void_p, i64p, C.c_int32]),
    "batch_summary_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "debug_last_ksw_mode": (C.c_int, []),
    "debug_last_ksw_pairs": (C.c_int, []),
    "debug_ksw_duo_batch": (C.c_int, [C.c_int32, u8p, i64p, u8p, i64p, u8p, i64p, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, u32p, C.c_int32]),
    "debug_limits": (C.c_int, [i64p]),
    "debug_set": (C.c_int, [C.c_char_p, C.c_int64]),
    "ksw_last_status": (C.c_int, []),
    # multi-GPU (dist_host.h): the end-of-job gather over librccl behind the C ABI
    "dist_unique_id": (C.c_int, [C.c_void_p, C.c_int64]),
    "dist_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "dist_rank": (C.c_int, [C.c_void_p]),
    "dist_world": (C.c_int, [C.c_void_p]),
    "dist_gather_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, i64p, C.c_void_p, C.c_int64, i64p, i64p]),
    "dist_gather_summaries": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, i64p, C.c_void_p, C.c_int64, i64p, i64p]),
    "dist_gather_payload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(BatchOut), i64p]),
    "dist_finalize": (C.c_int, [C.c_void_p]),
}
IHP_DIST_ID_BYTES = 128

KSW_ARGTYPES = [C.c_void_p, C.c_int, u8p, C.c_int, u8p, C.c_int8, i8p, C.c_int8, C.c_int8,
                C.c_int, C.c_int, C.c_int, C.POINTER(KswExtz)]

PRODUCT_SYMBOLS = ["ihp_" + n for n in list(_COMMON) + list(_PRODUCT_ONLY)] + ["ksw_extz2_sse"]


class Bound:
    """Function table of one loaded library (attribute access without prefix)."""

    def __init__(self, cdll, prefix, product):
        self.cdll, self.prefix = cdll, prefix
        table = dict(_COMMON)
        if product:
            table.update(_PRODUCT_ONLY)
        for name, (res, args) in table.items():
            fn = getattr(cdll, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


def bind(cdll, prefix="ihp_", product=True):
    return Bound(cdll, prefix, product)


def ptr(a, typ):
    """ctypes pointer to a C-contiguous numpy array (or NULL for None)."""
    if a is None:
        return C.cast(None, typ)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(typ)
