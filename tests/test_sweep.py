"""From decoded reads to VCF lines: gen_roi (f4) -> staging/batching (f3) -> assemble/ksw2/tally(/fallback) ->
filters and Variant records (f2), on a small synthetic chromosome with planted indels."""
import numpy as np
import pytest

from indelope_amd import sweep

OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def words(parts):
    return np.array([n << 4 | OPS[o] for n, o in parts if n > 0 or o == "M"], np.uint32)


def make_target(seed=3, length=60_000, every=4000, n_per_site=40, read_len=150):
    """A random chromosome with an indel planted every `every` bp; reads drawn around each site from both haplotypes,
    with the CIGAR an aligner would report (one I or D), plus background reads without events."""
    rng = np.random.default_rng(seed)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), length).tobytes()
    truth, reads, cigars = [], [], []
    for p in range(every, length - every, every):
        kind = "D" if rng.random() < 0.5 else "I"
        ell = int(rng.integers(5, 30))
        ins = rng.choice(np.frombuffer(b"ACGT", np.uint8), ell).tobytes()
        truth.append((p, kind, ell, ins))
        for _ in range(n_per_site):
            s = int(rng.integers(p - read_len + 25, p - 25))
            if rng.random() < 0.5:                                         # reference haplotype
                seq, cg, stop = ref[s:s + read_len], words([(read_len, "M")]), s + read_len
            elif kind == "D":
                a = p - s
                seq = ref[s:p] + ref[p + ell:p + ell + read_len - a]
                cg, stop = words([(a, "M"), (ell, "D"), (read_len - a, "M")]), s + read_len + ell
            else:
                a = p - s
                tail = read_len - a - ell
                seq = ref[s:p] + ins + ref[p:p + max(tail, 0)]
                seq = seq[:read_len]
                k = min(ell, read_len - a)
                cg, stop = words([(a, "M"), (k, "I"), (max(tail, 0), "M")]), s + a + max(tail, 0)
            reads.append(sweep.Read(seq, None, s, stop, 60))
            cigars.append(cg)
    order = np.argsort([r.start for r in reads], kind="stable")
    return ref, [reads[i] for i in order], [cigars[i] for i in order], truth


def run(api, batch_regions):
    ref, reads, cigars, truth = make_target()
    p = api.params(min_reads=3, min_ctg_len=73)                            # CLI defaults, indelope.nim:568-570
    lines, rois = sweep.call_target(api, reads, cigars, lambda a, b: ref[a:b], p, batch_regions=batch_regions,
                                    target_len=len(ref))
    return lines, rois, truth, ref


def check(lines, rois, truth, ref):
    assert len(rois) >= 0.9 * len(truth)
    ref = ref.decode()
    calls = {}
    for ln in lines:
        f = ln.split("\t")
        calls[int(f[1])] = (f[3], f[4], f[9].split(":")[0])
    hit = 0
    for p, kind, ell, ins in truth:
        for pos, (r, a, gt) in calls.items():
            if abs(pos - p) > 40 or gt != "0/1":
                continue
            if kind == "D" and len(r) == ell + 1 and len(a) == 1 and ref[pos - 1:pos + ell] == r:
                hit += 1
                break
            if kind == "I" and len(a) == ell + 1 and len(r) == 1 and ref[pos - 1:pos] == r:
                hit += 1
                break
    assert hit >= 0.8 * len(truth), (hit, len(truth), len(lines))


def test_decoded_reads_to_vcf_lines_on_the_oracle(oracle):
    lines, rois, truth, ref = run(oracle, 10_000)
    check(lines, rois, truth, ref)
    assert run(oracle, 3)[0] == lines                                      # flush size does not matter


@pytest.mark.gpu
def test_decoded_reads_to_vcf_lines_on_the_device(hip, oracle):
    lines, rois, truth, ref = run(hip, 10_000)
    check(lines, rois, truth, ref)
    exp = run(oracle, 10_000)
    assert lines == exp[0] and rois == exp[1]
    assert run(hip, 5)[0] == lines
