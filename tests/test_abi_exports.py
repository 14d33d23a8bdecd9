"""The C-ABI library loads without a GPU and exports every symbol include/indelope_hip.h declares."""
import ctypes as C
import os
import re

import indelope_amd
from indelope_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "indelope_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(ihp_[a-z0-9_]+)\s*\(", src))
    names.add("ksw_extz2_sse")
    return names


def test_library_exports_every_declared_symbol():
    lib = indelope_amd.load_library()
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in sorted(decl):
        assert hasattr(lib, name), "missing export: " + name
    assert set(A.PRODUCT_SYMBOLS) <= decl, set(A.PRODUCT_SYMBOLS) - decl


def test_struct_layouts_match_header():
    assert C.sizeof(A.KswExtz) == 48            # ksw2.h:22-30 on LP64
    assert C.sizeof(A.Ez) == 40
    assert C.sizeof(A.Event) == 168 and A.EVENT_DTYPE.itemsize == 168
    assert C.sizeof(A.Params) == 120
    lib = indelope_amd.load_library()
    p = A.Params()
    lib.ihp_params_default.argtypes = [C.POINTER(A.Params)]
    lib.ihp_params_default(C.byref(p))
    assert p.struct_size == C.sizeof(A.Params)
    assert (p.K, p.bw, p.zdrop, p.min_reads, p.min_ctg_len, p.combine_min_overlap) == (27, 50, 400, 4, 74, 65)
    assert (p.match, p.mismatch, p.gap_open, p.gap_ext) == (1, -2, 4, 1)
    assert abs(p.min_overlap_pct - 0.88) < 1e-15 and abs(p.error - 1e-3) < 1e-18
    # alignment fallback: new_ez(mismatch=-2, gap_open=5, gap_ext=1) indelope.nim:318-319, align_to defaults ksw2.nim:159
    assert (p.fallback, p.fb_match, p.fb_mismatch, p.fb_gap_open, p.fb_gap_ext) == (1, 1, -2, 5, 1)
    assert (p.fb_bw, p.fb_zdrop, p.fb_flag) == (-1, -1, 0)


def test_host_helpers_need_no_gpu(oracle):
    """encode/matrix/genotype are host code in the product library too."""
    api = indelope_amd.api()
    assert api.encode("ACGTNacgtn").tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4]
    assert api.matrix().tolist() == oracle.matrix().tolist()
    import kats
    kats.kat_genotype(api)
    for r, a, e in [(3, 9, 1e-3), (0, 5, 1e-3), (40, 2, 1e-2)]:
        g, o = api.genotype(r, a, e), oracle.genotype(r, a, e)
        assert g.gt == o.gt and list(g.gl) == list(o.gl) and api.qual(g) == oracle.qual(o)


def _build_harness():
    import subprocess
    lib_dir = os.path.join(ROOT, "indelope_amd", "lib")
    exe = os.path.join(ROOT, "tests", "abi_harness.bin")
    src = os.path.join(ROOT, "tests", "abi_harness.c")
    hdr = os.path.join(ROOT, "include", "indelope_hip.h")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        indelope_amd.load_library()                      # the library must have been built
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", lib_dir,
                               "-lindelope_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_harness_sees_the_same_layouts():
    """tests/abi_harness.c compiled by gcc against include/indelope_hip.h and linked with the library: struct sizes and
    field offsets as a C compiler lays them out equal the ctypes mirrors, and the host-only entry points answer."""
    import json
    import subprocess
    out = json.loads(subprocess.run([_build_harness()], capture_output=True, text=True, check=True).stdout)
    pairs = {"ksw_extz_t": A.KswExtz, "ihp_ez": A.Ez, "ihp_contig": A.Contig, "ihp_correction": A.Correction, "ihp_match": A.Match,
             "ihp_genotype_t": A.Genotype, "ihp_params": A.Params, "ihp_batch_in": A.BatchIn, "ihp_event": A.Event,
             "ihp_batch_out": A.BatchOut, "ihp_variant": A.Variant, "ihp_variants": A.Variants, "ihp_roi_in": A.RoiIn,
             "ihp_roi_out": A.RoiOut, "ihp_region_summary": A.RegionSummary}
    for cname, ct in pairs.items():
        assert out["sizeof_" + cname] == C.sizeof(ct), cname
    n = 0
    for key, off in out.items():
        if "." not in key or key.startswith("sizeof"):
            continue
        cname, field = key.split(".")
        assert getattr(pairs[cname], field).offset == off, key
        n += 1
    assert n >= 40
    assert out["params_default"] == [C.sizeof(A.Params), 27, 50, 400, 4, 74, 65, 5]
    assert out["matrix"] == [1, -2, -2, -2, 0, -2, 1, -2, -2, 0, -2, -2, 1, -2, 0, -2, -2, -2, 1, 0, 0, 0, 0, 0, 0]
    assert out["encode"] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4]
    assert (out["genotype_10_10"], out["genotype_0_0"]) == (A.IHP_GT_HET, A.IHP_GT_UNKNOWN)
    assert out["strerror_capacity"] == "capacity exceeded"
