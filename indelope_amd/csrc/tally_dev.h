// tally_dev.h -- events of one aligned contig, ref/alt k-mer choice and the
// per-read canonical k-mer tally (reference: src/ksw2/ksw2.nim:22-33,71-91;
// src/indelope.nim:229-311; third-party `kmer` package for mincode/dists --
// un-vendored and unpinned, so only upper-case ACGT semantics are claimed).
//
// One wavefront per aligned contig.  Event extraction and k-mer selection are
// wave-uniform scalar work; the tally gives each lane one read and rolls the
// forward / reverse-complement 2-bit codes along it (indelope.nim:300), then
// ballots the per-read found flags into ref/alt/both counts.
#pragma once
#include "ihp_common.h"

namespace ihp {

__device__ __forceinline__ int base2(uint8_t c)
{
	return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// canonical 2-bit code; false if a byte is not upper-case ACGT
__device__ inline bool mincode_dev(const char *kmer, int K, unsigned long long &code)
{
	unsigned long long f = 0, rc = 0;
	for (int i = 0; i < K; ++i) {
		const int b = base2((uint8_t)kmer[i]);
		if (b < 0) return false;
		f = (f << 2) | (unsigned long long)b;
		rc |= (unsigned long long)(3 - b) << (2 * i);
	}
	code = f < rc ? f : rc;
	return true;
}

// indelope.nim:293-311: lanes = reads.  counts[0..3) = ref_support, alt_support, both_found
__device__ inline void tally_reads(const uint8_t *bases, const long long *read_off, const uint8_t *mapq,
                                   long long r0, long long r1, int min_mapq, int K,
                                   unsigned long long refe, unsigned long long alte, int counts[3])
{
	const int lane = lane_id();
	const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1) : ~0ull;
	int nref = 0, nalt = 0, nboth = 0;
	for (long long b = r0; b < r1; b += 64) {
		const long long ri = b + lane;
		bool rf = false, af = false;
		if (ri < r1 && !(mapq && mapq[ri] < min_mapq)) {     // :294
			const uint8_t *seq = bases + read_off[ri];
			const int n = (int)(read_off[ri + 1] - read_off[ri]);
			unsigned long long f = 0, rc = 0;
			int valid = 0;
			for (int i = 0; i < n; ++i) {
				const int c = base2(seq[i]);
				if (c < 0) { valid = 0; f = rc = 0; continue; }
				f = ((f << 2) | (unsigned long long)c) & mask;
				rc = (rc >> 2) | ((unsigned long long)(3 - c) << (2 * (K - 1)));
				if (++valid < K) continue;
				const unsigned long long e = f < rc ? f : rc;
				rf |= e == refe;                                 // :301-309
				af |= e == alte;
			}
		}
		nref += popc64(ballot(rf));
		nalt += popc64(ballot(af));
		nboth += popc64(ballot(rf && af));                       // :310-311
	}
	counts[0] = nref; counts[1] = nalt; counts[2] = nboth;
}

__device__ __forceinline__ int distinct_bytes(const char *s, int n)
{
	unsigned seen_lo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int d = 0;
	for (int i = 0; i < n; ++i) {
		const unsigned c = (uint8_t)s[i];
		if (!((seen_lo[c >> 5] >> (c & 31)) & 1)) { seen_lo[c >> 5] |= 1u << (c & 31); d++; }
	}
	return d;
}

__device__ __forceinline__ bool same_bytes(const char *a, const char *b, int n)
{
	for (int i = 0; i < n; ++i) if (a[i] != b[i]) return false;
	return true;
}

struct TallyParams { int K, min_event_len, max_events, min_mapq_tally; };

// Ez.cigar truncation (ksw2.nim:22-33): number of events (I/D ops) among the ops the
// iterator yields; *ntrunc = number of ops yielded.
__device__ __forceinline__ int count_events(const uint32_t *cigar, int n_cigar, int max_q, int *ntrunc)
{
	const uint32_t max_off = (uint32_t)max_q;
	uint32_t off = 0;
	int nev = 0, nt = 0;
	for (int i = 0; i < n_cigar; ++i) {
		if (off >= max_off) break;
		const uint32_t op = cigar[i] & 0xf, len = cigar[i] >> 4;
		if (op != 2) off += len;
		nt++;
		if (op == 1 || op == 2) nev++;
	}
	*ntrunc = nt;
	return nev;
}

// Events of one alignment (the caller has checked 0 < nev <= max_events, indelope.nim:229).
// ctg_rel = ctg.start - region origin; reference points at the window the contig was
// aligned to (length reflen).  Lane 0 stores the nev events to ev[].
__device__ inline void fill_events(const uint32_t *cigar, int ntrunc,
                                   const uint8_t *ctg, int ctg_len, int ctg_rel,
                                   const uint8_t *reference, int reflen,
                                   const uint8_t *bases, const long long *read_off, const uint8_t *mapq,
                                   long long r0, long long r1, const TallyParams P, DevEvent *ev)
{
	const int lane = lane_id();
	const int K = P.K;
	const int width = (int)((double)(K + 1) / 2.0 - 1.0);                // :218
	int toff = ctg_rel, qoff = 0, ii = -1;
	for (int i = 0; i < ntrunc; ++i) {
		const uint32_t op = cigar[i] & 0xf, len = cigar[i] >> 4;
		if (op == 0) { toff += (int)len; qoff += (int)len; continue; }
		++ii;
		DevEvent E;
		E.len = len; E.pad = 0; E.fallback = 0; E.cf_offset = 0;
		E.ref_support = E.alt_support = E.both_found = 0;
		for (int k = 0; k < 32; ++k) E.ref_kmer[k] = E.alt_kmer[k] = 0;
		if (op == 1) {                                                   // ksw2.nim:75-76, :88-89
			E.type = 0; E.tstart_rel = toff; E.tstop_rel = toff + 1; E.qstart = qoff; E.qstop = qoff + (int)len;
			qoff += (int)len;
		} else {                                                         // ksw2.nim:77-78, :86-87
			E.type = 1; E.tstart_rel = toff; E.tstop_rel = toff + (int)len; E.qstart = qoff; E.qstop = qoff + 1;
			toff += (int)len;
		}
		int status = -1;
		unsigned long long refe = 0, alte = 0;
		if ((int)len < P.min_event_len) status = IHP_EV_SHORT;           // :234
		else if (reflen < K || ctg_len < K) status = IHP_EV_OOB;
		else {
			int tstart = E.tstart_rel - ctg_rel - width;                 // :236-238
			if (tstart < 0) tstart = 0;
			if (tstart + K > reflen) tstart = reflen - K;
			for (int k = 0; k < K; ++k) E.ref_kmer[k] = (char)reference[tstart + k];   // :240
			const int o1 = E.qstart, o2 = ctg_len - E.qstop - 1;
			E.cf_offset = o1 < o2 ? o1 : o2;                             // :243
			int qstart = E.qstart - width;                               // :244-246
			if (qstart < 0) qstart = 0;
			if (qstart + K > ctg_len) qstart = ctg_len - K;
			for (int k = 0; k < K; ++k) E.alt_kmer[k] = (char)ctg[qstart + k];          // :248
			if (same_bytes(E.alt_kmer, E.ref_kmer, K)) {                 // :255-262
				qstart = E.qstart - 3;
				if (qstart < 0) qstart = 0;
				if (qstart + K > ctg_len) {
					const int qend = E.qstop + 4 < ctg_len ? E.qstop + 4 : ctg_len;
					if (qend - K < 0) status = IHP_EV_OOB;
					else for (int k = 0; k < K; ++k) E.alt_kmer[k] = (char)ctg[qend - K + k];
				} else {
					for (int k = 0; k < K; ++k) E.alt_kmer[k] = (char)ctg[qstart + k];
				}
			}
			if (status < 0) {
				const bool same = same_bytes(E.alt_kmer, E.ref_kmer, K);
				if (same && (E.qstart == 0 || distinct_bytes(E.alt_kmer, K) == 1)) status = IHP_EV_SAME_KMER;   // :264
				else if (distinct_bytes(E.ref_kmer, K) < 3) status = IHP_EV_LOW_CPLX;                           // :266
				else if (same) status = IHP_EV_BUG_SAME;                                                       // :268-275
				else if (!mincode_dev(E.ref_kmer, K, refe) || !mincode_dev(E.alt_kmer, K, alte)) status = IHP_EV_NON_ACGT;
			}
		}
		if (status < 0) {
			int counts[3];
			tally_reads(bases, read_off, mapq, r0, r1, P.min_mapq_tally, K, refe, alte, counts);
			E.ref_support = counts[0]; E.alt_support = counts[1]; E.both_found = counts[2];
			E.fallback = counts[2] > 0;                                  // :313
			status = IHP_EV_TALLIED;
		}
		E.status = (unsigned char)status;
		if (lane == 0) ev[ii] = E;
	}
}

}  // namespace ihp
