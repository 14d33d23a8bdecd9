"""Build container only: regions of 257-600 reads (the `deep` workload, indelope_amd/synth.py; gen_roi hands over up to 600 reads
per roi, src/indelope.nim:483-485, :515) through BOTH restatements -- the C oracle and the Python transcription of the Nim
sources (oracle/nim_transcript.py) -- region by region, and 300 of the compared regions as a fixture for the CPU and GPU suites.

    python tests/golden/make_deep_golden.py [n_regions] [workers]

The fixture (tests/golden/deep_golden.npz) holds the expected flat results and, instead of 19 MB of read bases, the generator's
arguments with a SHA-256 of every input array it has to reproduce (indelope_amd/csrc/synth.cpp is part of this repository; a
generator that drifted is caught by the hash, not by a wrong expectation).
"""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
HERE = os.path.dirname(os.path.abspath(__file__))

from indelope_amd import synth                    # noqa: E402
from indelope_amd.host import BatchResult         # noqa: E402

IN_FIELDS = ("region_read_off", "read_off", "bases", "quals", "read_start", "read_stop", "mapq", "read_skip", "ref_off", "ref_bases", "ref_origin")
SETS = [("deep", 0, 220), ("deep", 5000, 40), ("deep_1e3", 0, 40)]      # (config, first region, regions): 300 regions in the fixture


def input_hash(b):
    h = hashlib.sha256()
    for f in IN_FIELDS:
        h.update(np.ascontiguousarray(getattr(b, f)).tobytes())
    return h.hexdigest()


def job(args):
    import make_transcript_golden as M
    import oracle as orc
    from oracle import nim_transcript as T
    name, first, n = args
    o = orc.get()
    o.use_reference_ksw(True)
    lib = T.load_reference_ksw2(M.REF_SO)
    b, _ = synth.config(name, n_regions=n, first_region=first)
    res = o.run_regions(b, o.params(K=27))
    diffs = []
    for r in range(b.n_regions):
        r0, r1 = int(b.region_read_off[r]), int(b.region_read_off[r + 1])
        reads = []
        for i in range(r0, r1):
            o0, o1 = int(b.read_off[i]), int(b.read_off[i + 1])
            reads.append(T.Read(b.bases[o0:o1].tobytes(), b.quals[o0:o1].tolist(), int(b.read_start[i]), int(b.read_stop[i]), int(b.mapq[i]), int(b.read_skip[i])))
        fai = T.Fai(b.ref_bases[int(b.ref_off[r]):int(b.ref_off[r + 1])].tobytes(), int(b.ref_origin[r]))
        mine = T.callsemble(reads, fai, lib, K=27, min_reads=4, min_ctg_len=74)
        d = M.compare(mine, M.oracle_region_record(res, r, b))
        if d:
            diffs.append((name, first + r, d))
    return name, first, n, diffs, int(res.ctg_support.max()), int((res.n_contigs_pre > 20).sum()), int((res.n_contigs_pre > 64).sum()), int(res.n_events)


def main():
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    jobs = []
    for name, first, n in SETS:                                             # the fixture's regions, in chunks
        for k in range(0, n, 20):
            jobs.append((name, first + k, min(20, n - k)))
    extra = max(0, n_total - sum(n for _, _, n in SETS))
    for k in range(0, extra, 20):                                           # more of both error rates, compared only
        jobs.append(("deep" if (k // 20) % 3 else "deep_1e3", 9000 + k, 20))
    tot, all_diffs, smax, g20, g64, ev = 0, [], 0, 0, 0, 0
    with mp.Pool(workers) as pool:
        for name, first, n, diffs, sm, a, b_, e in pool.imap_unordered(job, jobs):
            tot += n; all_diffs += diffs; smax = max(smax, sm); g20 += a; g64 += b_; ev += e
            print("%-9s regions %d..%d: %d differences" % (name, first, first + n, len(diffs)), flush=True)
    print("TOTAL %d regions of 257-600 reads, differences %d; largest support %d, regions with > 20 pre-combine contigs %d, > 64 %d, events %d"
          % (tot, len(all_diffs), smax, g20, g64, ev))
    for d in all_diffs[:20]:
        print("DIFF", d)
    if all_diffs:
        return 1
    import oracle as orc
    o = orc.get()
    o.use_reference_ksw(True)
    out = {}
    for name, first, n in SETS:
        b, _ = synth.config(name, n_regions=n, first_region=first)
        res = o.run_regions(b, o.params(K=27))
        key = "%s_%d_%d" % (name, first, n)
        out[key + ".sha256"] = np.frombuffer(input_hash(b).encode(), np.uint8)
        for f in BatchResult.FIELDS:
            out["%s.out.%s" % (key, f)] = getattr(res, f)
    np.savez_compressed(os.path.join(HERE, "deep_golden.npz"), **out)
    print("wrote deep_golden.npz: %d regions in %d sets" % (sum(n for _, _, n in SETS), len(SETS)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
