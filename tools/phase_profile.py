"""Per-phase shader cycles of the assembly kernels at steady state (one launch chain over N regions of a config, ihp_debug_set
profile): where a region's chain in k_asm_combine3 and a read's in k_asm_reads spend their cycles.
    python tools/phase_profile.py [config] [regions]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelope_amd
from indelope_amd import synth


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    api = indelope_amd.api()
    api.init(0)
    b, _ = synth.config(cfg, n_regions=n)
    b = b.with_trim_bounds()
    api.debug_set(profile=1)
    h = api.batch_upload(b)
    try:
        for _ in range(5):                       # the launch plan settles (CLEAN_MIN runs), then the one that counts
            api.batch_run(h); api.batch_sync(h)
        p0 = np.array(api.batch_profile(h), dtype=np.int64)
        api.batch_run(h); api.batch_sync(h)
        p = np.array(api.batch_profile(h), dtype=np.int64)
    finally:
        api.batch_free(h)
        api.debug_set(profile=0)
    d = p - p0 if (p >= p0).all() and p0[3] > 0 and p[3] > p0[3] else p
    R = max(int(d[3]), 1)
    nreads = int(b.region_read_off[-1])
    k = lambda v: "%8.1f k" % (v / R / 1e3)
    print("config %s, %d regions (%d through k_asm_combine3 in the run measured), cycles per region:" % (cfg, n, R))
    print("  combine: region total     ", k(d[2]))
    print("    take-over               ", k(d[0] - d[1]))
    print("    the two passes          ", k(d[1]))
    print("      best_match (exact)    ", k(d[5]), "  target offsets", k(d[49]), " query offsets", k(d[50]), " verify", k(d[51]))
    print("      vote scans            ", k(d[4]))
    print("      inserts               ", k(d[6]))
    print("      trims                 ", k(d[7]))
    print("      other                 ", k(d[1] - d[5] - d[4] - d[6] - d[7]))
    print("    epilogue                ", k(d[2] - d[0]))
    print("    by contigs at the take-over: >= 19: %d regions, %.1f k cycles each; 13-18: %d, %.1f k; <= 12: %d, %.1f k; the longest region of the launch %.1f k (region %d)" %
          (int(d[57]), d[54] / max(int(d[57]), 1) / 1e3, int(d[58]), d[55] / max(int(d[58]), 1) / 1e3, int(d[59]), d[56] / max(int(d[59]), 1) / 1e3, (int(p[53]) >> 20) / 1e3, int(p[53]) & 0xfffff))
    print("  events per region: best_match %.1f, candidates %.1f, verify passes %.1f, vote scans %.1f, merges %.1f" % tuple(d[32 + i] / R for i in range(5)))
    nj = max(int(d[18]) + int(d[19]), 1)
    print("  tally: %d jobs with events (%d without); cycles per job with events %.1f k, of which the header %.1f k" %
          (int(d[18]), int(d[19]), d[16] / max(int(d[18]), 1) / 1e3, d[20] / nj / 1e3))
    na = max(int(d[11]), 1)
    print("  ksw2 pair sweep, cycles per alignment (%d): set-up %.1f k, sweep %.1f k (early diagonals %.1f k, tail %.1f k), traceback %.1f k" %
          (na, d[8] / na / 1e3, d[9] / na / 1e3, d[60] / na / 1e3, d[61] / na / 1e3, d[10] / na / 1e3))
    print("  reads (per region, %.1f reads): set-up %s prep %s target %s query %s insert %s" %
          (nreads / n, k(d[27] * R / n), k(d[12] * R / n), k(d[13] * R / n), k(d[14] * R / n), k(d[15] * R / n)))


if __name__ == "__main__":
    main()
