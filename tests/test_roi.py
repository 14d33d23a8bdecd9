"""Row f4: the ROI evidence scan -- event_locations, gen_roi_internal, gen_roi (indelope.nim:430-545).

The oracle keeps the reference's sequential shape (a read cache flushed at coverage gaps); the device version works
on the whole run of reads at once.  Parity unpinned (no reference test covers gen_roi)."""
import numpy as np
import pytest

OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def cig(s):
    """'50M2D30M' -> BAM uint32 words."""
    import re
    return np.array([int(n) << 4 | OPS[o] for n, o in re.findall(r"(\d+)([MIDNSHP=X])", s)], np.uint32)


def ref_len(c):
    return int(sum(int(w >> 4) for w in c if (w & 0xf) in (0, 2, 3, 7, 8)))


def random_reads(rng, n, span, gap_every=0, skip_frac=0.05, event_frac=0.3, hot=()):
    """Reads in BAM order with random CIGARs; `hot`: positions where many reads carry the same deletion."""
    starts = np.sort(rng.integers(0, span - 200, n))
    if gap_every:
        starts = starts + (np.arange(n) // gap_every) * 400           # coverage gaps
    cigars, stops = [], []
    for s in starts:
        parts = []
        if rng.random() < 0.1:
            parts.append("%dS" % rng.integers(1, 20))
        left = int(rng.integers(60, 150))
        h = [p for p in hot if s + 10 < p < s + left - 10]
        if h:
            a = h[0] - s
            parts += ["%dM" % a, "7D", "%dM" % (left - a)]
        elif rng.random() < event_frac:
            a = int(rng.integers(5, left - 5))
            ev = rng.choice(["%dI", "%dD", "%dN", "%dX", "%d="]) % rng.integers(1, 12)
            parts += ["%dM" % a, ev, "%dM" % (left - a)]
        else:
            parts.append("%dM" % left)
        if rng.random() < 0.1:
            parts.append(rng.choice(["%dS", "%dI", "%dH"]) % rng.integers(1, 15))
        c = cig("".join(parts))
        cigars.append(c)
        stops.append(s + ref_len(c))
    skip = (rng.random(n) < skip_frac).astype(np.uint8)
    return starts.astype(np.int64), np.array(stops, np.int64), cigars, skip


def test_oracle_known_regions(oracle):
    """Five reads share a 3 bp deletion at 1050-1052, a sixth lies beyond a coverage gap with its own soft clip."""
    st = np.array([1000, 1010, 1020, 1030, 1040, 5000], np.int64)
    cg = [cig("%dM3D50M" % (1050 - s)) for s in st[:5]] + [cig("5S80M")]
    en = np.array([s + ref_len(c) for s, c in zip(st, cg)], np.int64)
    got = oracle.gen_roi(st, en, cg, min_event_support=4, min_read_coverage=4)
    assert got == [(1050, 1052, [0, 1, 2, 3, 4])]
    # evidence below the threshold, or too few / too many reads: nothing
    assert oracle.gen_roi(st, en, cg, min_event_support=6) == []
    assert oracle.gen_roi(st, en, cg, min_event_support=4, min_read_coverage=6) == []
    assert oracle.gen_roi(st, en, cg, min_event_support=4, min_read_coverage=2, max_read_coverage=4) == []
    # a skippable read adds no evidence and is not collected
    skip = np.array([0, 1, 0, 0, 0, 0], np.uint8)
    assert oracle.gen_roi(st, en, cg, read_skip=skip, min_event_support=4, min_read_coverage=4) == [(1050, 1052, [0, 2, 3, 4])]
    # ops that do not consume the reference mark one position; '=' / 'X' / 'N' count as events over their length
    cg2 = [cig("30M4I20M"), cig("30M2X18M"), cig("30M1I20M"), cig("30M3N17M")]
    st2 = np.array([100, 100, 100, 100], np.int64)
    en2 = np.array([s + ref_len(c) for s, c in zip(st2, cg2)], np.int64)
    assert oracle.gen_roi(st2, en2, cg2, min_event_support=4, min_read_coverage=1) == [(130, 130, [0, 1, 2, 3])]
    assert oracle.gen_roi(st2, en2, cg2, min_event_support=2, min_read_coverage=1) == [(130, 131, [0, 1, 2, 3])]


def test_oracle_windows_are_cut_where_the_cache_is_flushed(oracle):
    """A read that starts right after the cached reads end opens a new evidence window (indelope.nim:529-535): a run of
    evidence that touches both sides comes out as two regions."""
    st = np.array([100] * 4 + [181] * 4, np.int64)
    cg = [cig("80M5I")] * 4 + [cig("6S50M")] * 4              # evidence 4 at 180 (insertion at the end) and at 181 (clip)
    en = np.array([s + ref_len(c) for s, c in zip(st, cg)], np.int64)
    assert list(en[:4]) == [180] * 4
    got = oracle.gen_roi(st, en, cg, min_event_support=4, min_read_coverage=4)
    assert got == [(180, 180, [0, 1, 2, 3]), (181, 181, [4, 5, 6, 7])]
    # without the gap (one read spans both) there is one window and one region
    st2 = np.append(st, 150)
    order = np.argsort(st2, kind="stable")
    cg2 = [(cg + [cig("60M")])[i] for i in order]
    st2 = st2[order]
    en2 = np.array([s + ref_len(c) for s, c in zip(st2, cg2)], np.int64)
    got2 = oracle.gen_roi(st2, en2, cg2, min_event_support=4, min_read_coverage=4)
    assert [(a, b) for a, b, _ in got2] == [(180, 181)]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,span,gap,minev,minr,maxr", [
    (1, 3000, 20_000, 0, 4, 4, 600), (2, 5000, 60_000, 40, 3, 3, 600), (3, 800, 3_000, 0, 4, 4, 30),
    (4, 20_000, 400_000, 500, 4, 4, 600), (5, 2000, 5_000, 7, 2, 1, 600), (6, 300, 100_000, 3, 1, 1, 600),
    # min_event_support = 0: every position counts, so a window starts exactly where the cache was flushed -- at the start of the
    # FIRST read beyond every cached stop, skippable or not (tools/roi_stress.py found the device cutting again at the next read)
    (7, 400, 20_000, 3, 0, 1, 30), (8, 12_000, 150_000, 0, 0, 2, 100_000), (9, 60, 20_000, 3, 0, 1, 100_000)])
def test_device_scan_equals_oracle(hip, oracle, seed, n, span, gap, minev, minr, maxr):
    rng = np.random.default_rng(seed)
    hot = sorted(rng.integers(500, span - 500, max(3, span // 3000)).tolist())
    st, en, cg, skip = random_reads(rng, n, span, gap_every=gap, hot=hot)
    kw = dict(read_skip=skip, origin=0, span=int(en.max() + 50), min_event_support=minev, min_read_coverage=minr,
              max_read_coverage=maxr)
    exp = oracle.gen_roi(st, en, cg, **kw)
    got = hip.gen_roi(st, en, cg, **kw)
    assert len(exp) > 0
    assert got == exp
    # a non-zero origin shifts everything
    kw["origin"] = 1_000_000
    assert hip.gen_roi(st + 1_000_000, en + 1_000_000, cg, **kw) == [(a + 1_000_000, b + 1_000_000, r) for a, b, r in exp]


@pytest.mark.gpu
def test_device_scan_edges(hip, oracle):
    z = np.zeros(0, np.int64)
    assert hip.gen_roi(z, z, [], span=100) == [] == oracle.gen_roi(z, z, [], span=100)
    st = np.array([100] * 4 + [181] * 4, np.int64)
    cg = [cig("80M5I")] * 4 + [cig("6S50M")] * 4
    en = np.array([s + ref_len(c) for s, c in zip(st, cg)], np.int64)
    assert hip.gen_roi(st, en, cg) == oracle.gen_roi(st, en, cg) == [(180, 180, [0, 1, 2, 3]), (181, 181, [4, 5, 6, 7])]
    # evidence saturates at 255 (indelope.nim:541-543): 300 reads with the same deletion, threshold 255
    st = np.full(300, 10, np.int64)
    cg = [cig("20M2D20M")] * 300
    en = st + 42
    kw = dict(min_event_support=255, min_read_coverage=1, max_read_coverage=600)
    assert hip.gen_roi(st, en, cg, **kw) == oracle.gen_roi(st, en, cg, **kw) == [(30, 31, list(range(300)))]
    # a span shorter than the reads reach, all reads skippable, a single read: handled alike on both sides
    rng = np.random.default_rng(8)
    st2, en2, cg2, skip2 = random_reads(rng, 400, 6000, gap_every=25, hot=[900, 2500, 4100])
    for kw2 in (dict(span=3000), dict(span=int(en2.max()) - 7), dict(read_skip=np.ones(400, np.uint8)), dict(span=1)):
        kw2 = dict(dict(read_skip=skip2, min_event_support=3, min_read_coverage=2), **kw2)
        assert hip.gen_roi(st2, en2, cg2, **kw2) == oracle.gen_roi(st2, en2, cg2, **kw2), kw2
    one = (np.array([50], np.int64), np.array([150], np.int64), [cig("40M3D57M")])
    kw1 = dict(min_event_support=1, min_read_coverage=1)
    assert hip.gen_roi(*one, **kw1) == oracle.gen_roi(*one, **kw1) == [(90, 92, [0])]
    # more reads than max_read_coverage: the region is dropped (:483-485)
    kw["max_read_coverage"] = 299
    assert hip.gen_roi(st, en, cg, **kw) == oracle.gen_roi(st, en, cg, **kw) == []


@pytest.mark.gpu
def test_a_skippable_read_flushes_the_cache_once(hip, oracle):
    """indelope.nim:529-537 with min_event_support = 0: read B (skippable) lies beyond the cached read A and flushes the window;
    C, behind it, finds the cache empty and flushes nothing -- the next window starts at B's start, not at C's."""
    st = np.array([0, 200, 250], np.int64)
    en = np.array([100, 300, 350], np.int64)
    cg = [cig("100M"), cig("100M"), cig("100M")]
    skip = np.array([0, 1, 0], np.uint8)
    kw = dict(read_skip=skip, origin=0, span=400, min_event_support=0, min_read_coverage=1, max_read_coverage=10)
    exp = oracle.gen_roi(st, en, cg, **kw)
    assert [(a, b) for a, b, _ in exp] == [(0, 199), (200, 400)]
    assert hip.gen_roi(st, en, cg, **kw) == exp
