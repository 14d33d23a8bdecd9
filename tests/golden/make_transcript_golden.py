"""Build container only: the Python transcription of the Nim sources (oracle/nim_transcript.py, written without looking at
oracle/*.c) against the C oracle, region by region, over the parity parameter space -- and a few hundred of the compared
regions as fixtures for the GPU suite (tests/golden/transcript_golden.npz: inputs + the expected flat results, which BOTH
restatements produced).

    python tests/golden/make_transcript_golden.py [n_regions_total] [workers]

Workloads: tandem-repeat expansions (the `alt_kmer == ref_kmer` retry), the BASELINE shapes (C2 / C3 / C5-like), high error rates (many single-read contigs: > 20 pre-combine contigs,
votes firing in combine), tandem duplications (the alignment fallback), low base qualities at the read ends and reads that
empty under trim, low mapping qualities on either side of the three thresholds (5 / 10 / 20), skippable reads, the CLI's
parameters (min_reads 3, min_ctg_len 73) beside the proc defaults, K = 21 / 27 / 31.
Every difference is printed with the region's seed; the script exits 1 when there is one.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libksw2_ref.so")

from indelope_amd import synth                    # noqa: E402
from indelope_amd import _abi as A                # noqa: E402
from indelope_amd.host import BatchResult, RegionBatch   # noqa: E402

EV_WHERE = {A.IHP_EV_SHORT: 234, A.IHP_EV_SAME_KMER: 264, A.IHP_EV_LOW_CPLX: 266, A.IHP_EV_BUG_SAME: 268, A.IHP_EV_OOB: "oob",
            A.IHP_EV_NON_ACGT: "non-acgt", A.IHP_EV_TALLIED: "tallied"}


def workloads(n_total, seed=2025):
    """(name, generator kwargs, params kwargs, mutate) tuples whose region counts add up to about n_total."""
    rng = np.random.default_rng(seed)
    out = []
    shapes = [("c2", dict(read_len=150, n_reads=(64, 64), err_rate=1e-3), 0.16), ("c3", dict(read_len=150, n_reads=(16, 256), err_rate=1e-3), 0.08),
              ("c5", dict(read_len=300, n_reads=(64, 64), err_rate=1e-3, n_events=2, window_len=1400, event_pos=500), 0.04),
              ("clean", dict(read_len=150, n_reads=(8, 48), err_rate=0.0), 0.10), ("noisy", dict(read_len=150, n_reads=(20, 64), err_rate=1e-2), 0.14),
              ("verynoisy", dict(read_len=100, n_reads=(24, 48), err_rate=3e-2), 0.10), ("dup", dict(read_len=150, n_reads=(12, 48), err_rate=1e-3, dup_frac=0.7), 0.12),
              ("short", dict(read_len=80, n_reads=(10, 40), err_rate=2e-3), 0.08), ("lowq", dict(read_len=150, n_reads=(16, 64), err_rate=2e-3), 0.12),
              ("refmut", dict(read_len=150, n_reads=(16, 48), err_rate=1e-3), 0.10), ("repeat", dict(), 0.06)]
    cid = 7000
    for name, g, share in shapes:
        n = max(8, int(n_total * share))
        per = 40 if name not in ("c3", "c5") else 16
        for k in range(0, n, per):
            cid += 1
            K = int(rng.choice([21, 27, 27, 27, 31])) if name != "c5" else 31
            pk = dict(K=K)
            if rng.random() < 0.4:
                pk.update(min_reads=3, min_ctg_len=73)                       # the CLI's values (indelope.nim:568-570)
            out.append((name, dict(g, n_regions=min(per, n - k), config_id=cid), pk, name in ("lowq", "dup", "noisy") or rng.random() < 0.25))
    return out


def mutate(b, rng):
    """Low qualities at the read ends, reads that empty under trim, mapping qualities around 5 / 10 / 20, skippable reads."""
    q = b.quals.copy()
    mapq = b.mapq.copy()
    skip = b.read_skip.copy()
    for i in range(b.n_reads):
        o0, o1 = int(b.read_off[i]), int(b.read_off[i + 1])
        u = rng.random()
        if u < 0.25:
            q[o0:o0 + int(rng.integers(1, 12))] = rng.integers(2, 15)
        if 0.15 < u < 0.4:
            q[o1 - int(rng.integers(1, 12)):o1] = rng.integers(2, 15)
        if u > 0.985:
            q[o0:o1] = 2                                                     # emptied (trim returns high)
        if u > 0.97 and u <= 0.985:
            q[o0 + 1:o1] = 3                                                 # one base left ... which trim() also empties or keeps per :28-33
        v = rng.random()
        if v < 0.12:
            mapq[i] = int(rng.choice([0, 4, 5, 6, 9, 10, 11, 19, 20, 21]))
        if v > 0.985:
            skip[i] = 1
    return RegionBatch(b.region_read_off, b.read_off, b.bases, q, b.read_start, b.read_stop, mapq, skip, b.ref_off, b.ref_bases, b.ref_origin)


def mutate_ref(b, rng):
    """Edits of the reference windows only (the reads stay): 1-3 base indels (events below min_event_len, several events per
    alignment -- the `len(qlocs) > 4` exit), dinucleotide and homopolymer stretches around the planted event (the
    `ref_kmer.toSet.len < 3` and `alt_kmer == ref_kmer` exits)."""
    ref = b.ref_bases.copy()
    for r in range(b.n_regions):
        f0, f1 = int(b.ref_off[r]), int(b.ref_off[r + 1])
        w = ref[f0:f1].copy()
        L = len(w)
        u = rng.random()
        if u < 0.5:
            for _ in range(int(rng.integers(1, 6))):
                p, n = int(rng.integers(60, L - 80)), int(rng.integers(1, 4))
                if rng.random() < 0.5:
                    w = np.concatenate([w[:p], w[p + n:], rng.choice(np.frombuffer(b"ACGT", np.uint8), n)])
                else:
                    w = np.concatenate([w[:p], rng.choice(np.frombuffer(b"ACGT", np.uint8), n), w[p:]])[:L]
        elif u < 0.75:
            p = L // 2 - int(rng.integers(10, 50))
            w[p:p + 70] = np.frombuffer((b"AC" * 35), np.uint8)
        elif u < 0.9:
            p = L // 2 - int(rng.integers(10, 50))
            w[p:p + 60] = ord("T")
        ref[f0:f1] = w
    return RegionBatch(b.region_read_off, b.read_off, b.bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip, b.ref_off, ref, b.ref_origin)


def repeat_batch(n_regions, seed, read_len=150):
    """Regions whose event is an expansion / contraction of a tandem repeat or homopolymer longer than K: the alternate k-mer
    equals the reference k-mer at first (indelope.nim:255-262 retries further left or at the end), some stay equal (:264)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    rro, ro, bases, quals, rstart, rstop, mapq, skip, fo, ref, origin = [0], [0], [], [], [], [], [], [], [0], [], []
    for r in range(n_regions):
        unit = rng.choice(acgt, int(rng.choice([1, 1, 2, 3, 4])))
        m = int(rng.integers(40, 70)) // len(unit) + 8
        left, right = rng.choice(acgt, 220), rng.choice(acgt, 260)
        d = int(rng.integers(5, 30)) // len(unit) + 1
        if rng.random() < 0.5:
            d = -min(d, m - 6)
        W = np.concatenate([left, np.tile(unit, m), right])
        alt = np.concatenate([left, np.tile(unit, m + d), right])
        shift = d * len(unit)
        org = 1_000_000 + 10_000 * r
        reads = []
        for _ in range(int(rng.integers(16, 48))):
            hap, sh = (alt, shift) if rng.random() < 0.6 else (W, 0)
            st = int(rng.integers(60, 230))
            sq = hap[st:st + read_len].copy()
            err = rng.random(len(sq)) < 1e-3
            sq[err] = rng.choice(acgt, int(err.sum()))
            reads.append((st, sq))
        reads.sort(key=lambda x: x[0])                           # BAM order (stable: ties keep generation order)
        for st, sq in reads:
            bases.append(sq); quals.append(np.full(len(sq), 30, np.uint8))
            ro.append(ro[-1] + len(sq)); rstart.append(org + st); rstop.append(org + st + len(sq)); mapq.append(60); skip.append(0)
        rro.append(rro[-1] + len(reads))
        ref.append(W); fo.append(fo[-1] + len(W)); origin.append(org)
    return RegionBatch(np.array(rro, np.int64), np.array(ro, np.int64), np.concatenate(bases), np.concatenate(quals), np.array(rstart, np.int64),
                       np.array(rstop, np.int64), np.array(mapq, np.uint8), np.array(skip, np.uint8), np.array(fo, np.int64), np.concatenate(ref),
                       np.array(origin, np.int64))


def make_batch(name, g):
    if name == "repeat":
        return repeat_batch(g["n_regions"], g["config_id"])
    return synth.generate(**g)[0]


def oracle_region_record(res, r, b):
    """The C oracle's flat results of region r in the transcript's shape."""
    c0, c1 = int(res.contig_off[r]), int(res.contig_off[r + 1])
    nr = int(b.region_read_off[r + 1] - b.region_read_off[r])
    out = {"n_pre": int(res.n_contigs_pre[r]), "contigs": []}
    for c in range(c0, c1):
        s0, s1 = int(res.ctg_seq_off[c]), int(res.ctg_seq_off[c + 1])
        rec = {"start": int(res.ctg_start[c]), "nreads": int(res.ctg_nreads[c]), "seq": res.ctg_seq[s0:s1].tobytes(),
               "support": res.ctg_support[s0:s1].tolist(), "aligned": bool(res.aligned_flag(c)) if hasattr(res, "aligned_flag") else bool(res.aln_flags[c] & A.IHP_ALN_DONE)}
        if rec["aligned"]:
            ez = res.aln_ez[c]
            rec["ez"] = {k: int(ez[k]) for k in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar")}
            rec["clamped"] = bool(res.aln_flags[c] & A.IHP_ALN_REF_CLAMPED)
            rec["ref_len"] = int(res.aln_ref_len[c])
            w = res.cigar[int(res.cigar_off[c]):int(res.cigar_off[c + 1])].tolist()
            rec["full_cigar"] = [(x & 0xf, x >> 4) for x in w]
            rec["events"] = []
            for e in range(int(res.event_off[c]), int(res.event_off[c + 1])):
                E = res.events[e]
                ev = {"tstart": int(E["tstart"]), "tstop": int(E["tstop"]), "qstart": int(E["qstart"]), "qstop": int(E["qstop"]), "len": int(E["len"]),
                      "type": int(E["type"]), "where": EV_WHERE[int(E["status"])]}
                if ev["where"] in (264, 266, 268, "tallied", "non-acgt"):
                    ev.update(cf_offset=int(E["cf_offset"]), ref_kmer=bytes(E["ref_kmer"]), alt_kmer=bytes(E["alt_kmer"]))
                if ev["where"] == "tallied":
                    h0 = int(res.hit_off[e])
                    ev.update(kmer_ref_support=int(E["kmer_ref_support"]), kmer_alt_support=int(E["kmer_alt_support"]), kmer_both_found=int(E["kmer_both_found"]),
                              fallback_needed=bool(E["fallback_needed"]), aligned=bool(E["aligned"]), ref_support=int(E["ref_support"]),
                              alt_support=int(E["alt_support"]), both_found=int(E["both_found"]), gt=int(E["gt"]), gl=[float(x) for x in E["gl"]],
                              ref_hit=res.ref_hit[h0:h0 + nr].tolist(), alt_hit=res.alt_hit[h0:h0 + nr].tolist())
                rec["events"].append(ev)
        out["contigs"].append(rec)
    return out


def compare(mine, theirs):
    """None, or the first difference between the transcript's record and the oracle's."""
    if mine["n_pre"] != theirs["n_pre"]:
        return "n_pre %d != %d" % (mine["n_pre"], theirs["n_pre"])
    if len(mine["contigs"]) != len(theirs["contigs"]):
        return "final contigs %d != %d" % (len(mine["contigs"]), len(theirs["contigs"]))
    for k, (a, b) in enumerate(zip(mine["contigs"], theirs["contigs"])):
        for f in ("start", "nreads", "seq", "support", "aligned"):
            if a[f] != b[f]:
                return "contig %d %s: %r != %r" % (k, f, a[f] if f != "support" else a[f][:12], b[f] if f != "support" else b[f][:12])
        if not a["aligned"]:
            continue
        for f in ("ez", "clamped", "ref_len", "full_cigar"):
            if a[f] != b[f]:
                return "contig %d %s: %r != %r" % (k, f, a[f], b[f])
        ae, be = a["events"], b["events"]
        if "n_qlocs" in a:                                         # :229: nothing behind it; the oracle keeps no events either
            if be:
                return "contig %d: %d query events, the oracle kept %d" % (k, a["n_qlocs"], len(be))
            continue
        if len(ae) != len(be):
            return "contig %d events %d != %d" % (k, len(ae), len(be))
        for j, (x, y) in enumerate(zip(ae, be)):
            for f in y:
                if f == "gl":
                    if not np.allclose(x[f], y[f], rtol=1e-12, atol=0):
                        return "contig %d event %d gl %r != %r" % (k, j, x[f], y[f])
                    continue
                if f in ("ref_kmer", "alt_kmer"):
                    if x.get(f) != y[f].rstrip(b"\0"):
                        return "contig %d event %d %s: %r != %r" % (k, j, f, x.get(f), y[f])
                    continue
                if x.get(f) != y[f]:
                    return "contig %d event %d %s: %r != %r (where %r / %r)" % (k, j, f, x.get(f), y[f], x.get("where"), y.get("where"))
    return None


def job(args):
    name, g, pk, mut = args
    import oracle as orc
    from oracle import nim_transcript as T
    o = orc.get()
    o.use_reference_ksw(True)
    lib = T.load_reference_ksw2(REF_SO)
    b = make_batch(name, g)
    if mut:
        b = mutate(b, np.random.default_rng(g["config_id"]))
    if name == "refmut":
        b = mutate_ref(b, np.random.default_rng(g["config_id"] + 1))
    res = o.run_regions(b, o.params(**pk))
    diffs = []
    retried = 0
    for r in range(b.n_regions):
        r0, r1 = int(b.region_read_off[r]), int(b.region_read_off[r + 1])
        reads = []
        for i in range(r0, r1):
            o0, o1 = int(b.read_off[i]), int(b.read_off[i + 1])
            reads.append(T.Read(b.bases[o0:o1].tobytes(), b.quals[o0:o1].tolist(), int(b.read_start[i]), int(b.read_stop[i]), int(b.mapq[i]), int(b.read_skip[i])))
        fai = T.Fai(b.ref_bases[int(b.ref_off[r]):int(b.ref_off[r + 1])].tobytes(), int(b.ref_origin[r]))
        kw = dict(K=pk["K"], min_reads=pk.get("min_reads", 4), min_ctg_len=pk.get("min_ctg_len", 74))
        mine = T.callsemble(reads, fai, lib, **kw)
        retried += sum(1 for c in mine["contigs"] for e in c.get("events", []) if e.get("retried"))
        d = compare(mine, oracle_region_record(res, r, b))
        if d:
            diffs.append((name, g["config_id"], r, d))
    stats = dict(regions=b.n_regions, kmer_retry=retried, pre_gt20=int((res.n_contigs_pre > 20).sum()), contigs=int(res.n_contigs), events=int(res.n_events),
                 tallied=int((res.events["status"] == 0).sum()), fallback=int((res.events["aligned"] == 1).sum()),
                 same_kmer=int((res.events["status"] == A.IHP_EV_SAME_KMER).sum()), short=int((res.events["status"] == A.IHP_EV_SHORT).sum()),
                 low_cplx=int((res.events["status"] == A.IHP_EV_LOW_CPLX).sum()), clamped=int(((res.aln_flags & A.IHP_ALN_REF_CLAMPED) != 0).sum()),
                 aligned=int(((res.aln_flags & A.IHP_ALN_DONE) != 0).sum()),
                 trailing_d=int(sum(1 for c in range(res.n_contigs) if res.cigar_off[c + 1] > res.cigar_off[c] and (int(res.cigar[res.cigar_off[c + 1] - 1]) & 0xf) == 2)))
    return name, g, pk, mut, diffs, stats


def main():
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 5200
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    assert os.path.exists(REF_SO), "needs oracle/_ref (make -C oracle ref): build container only"
    wl = workloads(n_total)
    tot = {}
    all_diffs = []
    keep = []
    with mp.Pool(workers) as pool:
        for name, g, pk, mut, diffs, stats in pool.imap_unordered(job, wl):
            for k, v in stats.items():
                tot[k] = tot.get(k, 0) + v
            all_diffs += diffs
            if not diffs:
                keep.append((name, g, pk, mut))
            print("%-10s id %d: %d regions, %d differences" % (name, g["config_id"], stats["regions"], len(diffs)), flush=True)
    print("TOTAL", tot, "differences:", len(all_diffs))
    for d in all_diffs[:40]:
        print("DIFF", d)
    # fixtures: a few regions of every compared workload kind (inputs + expected flat results), ~300 regions in all
    import oracle as orc
    o = orc.get()
    o.use_reference_ksw(True)
    out, n_fix = {}, 0
    per_kind = {}
    for name, g, pk, mut in sorted(keep, key=lambda x: x[1]["config_id"]):
        if per_kind.get(name, 0) >= 3:
            continue
        per_kind[name] = per_kind.get(name, 0) + 1
        g2 = dict(g, n_regions=min(g["n_regions"], 12))
        b = make_batch(name, g2)
        if mut:
            b = mutate(b, np.random.default_rng(g["config_id"]))   # (the same stream as the compared batch: its first reads)
        if name == "refmut":
            b = mutate_ref(b, np.random.default_rng(g["config_id"] + 1))
        res = o.run_regions(b, o.params(**pk))
        key = "%s_%d" % (name, g["config_id"])
        for f in ("region_read_off", "read_off", "bases", "quals", "read_start", "read_stop", "mapq", "read_skip", "ref_off", "ref_bases", "ref_origin"):
            out["%s.in.%s" % (key, f)] = getattr(b, f)
        for f in BatchResult.FIELDS:
            out["%s.out.%s" % (key, f)] = getattr(res, f)
        out["%s.params" % key] = np.array([pk["K"], pk.get("min_reads", 4), pk.get("min_ctg_len", 74)], np.int32)
        n_fix += b.n_regions
    if not all_diffs:
        np.savez_compressed(os.path.join(HERE, "transcript_golden.npz"), **out)
        print("wrote transcript_golden.npz: %d regions in %d sets" % (n_fix, sum(per_kind.values())))
    return 1 if all_diffs else 0


if __name__ == "__main__":
    sys.exit(main())
