// contig_dev.h -- device side of the slide-assembly (reference: src/contig.nim).
//
// One wavefront owns one region.  Contig bases and supports live in a per-wave
// arena (byte array + u32 array indexed by the same element offset); contig
// metadata lives in LDS.  The reference's sequential scan over candidate offsets
// (contig.nim:86-139) becomes: lanes = candidate offsets, an 8-base prefilter per
// lane, then a wave-cooperative full compare of each surviving offset with
// ballot/popcount tallies.  Acceptance uses the reference's total order
// (matches desc, mismatches asc, contig index asc, scan order asc) so that the
// result is bit-identical to the serial scan.
#pragma once
#include "ihp_common.h"

namespace ihp {

constexpr int QSLOT = MAXC;      // metadata slot of the read currently being inserted

struct Corr { int qoff; int toff; int qbest; };   // contig.nim:17

struct RegionState {             // LDS, one per wave
	int off[MAXC + 1], len[MAXC + 1], cap[MAXC + 1];
	long long nreads[MAXC + 1], start[MAXC + 1];
	unsigned char alive[MAXC + 1];
	short listA[MAXC], listB[MAXC];
	unsigned bitmap[MAXLEN / 32];
	int bump;
	int err;
};

struct Arena {
	uint8_t *seq;                // [cap]
	uint32_t *sup;               // [cap]
	int cap;                     // total elements
	int stage_off;               // [stage_off, cap) is the staging area of the read being inserted
	Corr *corr;                  // [corr_cap] corrections of the winning offset
	int corr_cap;
};

struct Best {
	int found, ma, mm, pos, slot, off;
};

__device__ __forceinline__ long long wave_min_ll(long long v)
{
	for (int d = 32; d >= 1; d >>= 1) {
		long long o = __shfl_xor(v, d, 64);
		v = o < v ? o : v;
	}
	return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

// dst/src may overlap only with dst < src (ascending chunks; every lane's load
// feeds its own store, so a chunk is fully read before it is written).
__device__ __forceinline__ void wcopy(uint8_t *dseq, uint32_t *dsup, const uint8_t *sseq, const uint32_t *ssup, int n)
{
	const int lane = lane_id();
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		const bool in = i < n;
		uint8_t b = 0; uint32_t s = 0;
		if (in) { b = sseq[i]; s = ssup[i]; }
		WSYNC();
		if (in) { dseq[i] = b; dsup[i] = s; }
	}
	WSYNC();
}

__device__ __forceinline__ int alloc_slot(RegionState &S)
{
	const int lane = lane_id();
	for (int b = 0; b < MAXC; b += 64) {
		unsigned long long m = ballot(!S.alive[b + lane]);
		if (m) return b + ctz64(m);
	}
	return -1;
}

// Slide every live contig down to close the holes left by relocations.
__device__ inline void compact(RegionState &S, Arena &A)
{
	const int lane = lane_id();
	int newbump = 0;
	long long last = -1;
	for (;;) {
		long long key = 0x7fffffffffffffffll;
		for (int b = 0; b < MAXC; b += 64) {
			int s = b + lane;
			if (S.alive[s] && (long long)S.off[s] > last) {
				long long k = ((long long)S.off[s] << 16) | s;
				key = k < key ? k : key;
			}
		}
		key = wave_min_ll(key);
		if (key == 0x7fffffffffffffffll) break;
		int s = (int)(key & 0xffff), o = (int)(key >> 16), n = S.len[s];
		WSYNC();
		if (o != newbump) wcopy(A.seq + newbump, A.sup + newbump, A.seq + o, A.sup + o, n);
		if (lane == 0) { S.off[s] = newbump; S.cap[s] = n; }
		WSYNC();
		newbump += n;
		last = o;
	}
	if (lane == 0) S.bump = newbump;
	WSYNC();
}

__device__ inline bool ensure_space(RegionState &S, Arena &A, int need)
{
	if (S.bump + need <= A.stage_off) return true;
	compact(S, A);
	return S.bump + need <= A.stage_off;
}

__device__ __forceinline__ int headroom(int n) { int h = n >> 1; return h < 128 ? 128 : h; }

// ---- slide_align + best_match (contig.nim:70-141, :224-240) -----------------
// Scan one target contig; update `best` under the reference's total order.
__device__ inline void slide_scan(const RegionState &S, const Arena &A, int qs, int ts, int pos,
                                  int min_overlap, int max_mm, int rule, Best &best)
{
	const int lane = lane_id();
	const uint8_t *qseq = A.seq + S.off[qs], *tseq = A.seq + S.off[ts];
	const uint32_t *qsup = A.sup + S.off[qs], *tsup = A.sup + S.off[ts];
	const int qlen = S.len[qs], tlen = S.len[ts];
	const long long qreads = S.nreads[qs], treads = S.nreads[ts];
	const int omax = tlen - min_overlap;                       // :79
	int omin_abs = qlen - min_overlap;                         // :78, :114 abs(omin)
	if (omin_abs < 0) omin_abs = -omin_abs;
	const int n1 = omax >= 0 ? omax + 1 : 0;
	const int total = n1 + omin_abs;
	for (int base = 0; base < total; base += 64) {
		const int idx = base + lane;
		const bool active = idx < total;
		int qo0 = 0, to0 = 0;
		if (idx < n1) to0 = idx; else qo0 = idx - n1 + 1;
		int n = qlen - qo0 < tlen - to0 ? qlen - qo0 : tlen - to0;
		if (n < 0) n = 0;
		// an offset can only be accepted with ma >= min_overlap-1 (:81,:107) and only beats the
		// running best with ma >= best.ma; ma <= n bounds both.
		const int need = best.found && best.ma > min_overlap - 1 ? best.ma : min_overlap - 1;
		bool surv = active && n >= need;
		if (surv) {
			int mmf = 0;
			const int nf = n < FILTER_CH ? n : FILTER_CH;
			for (int k = 0; k < nf; ++k) {
				if (qseq[qo0 + k] != tseq[to0 + k] &&
				    !allowed(rule, qsup[qo0 + k], tsup[to0 + k], qreads, treads)) {
					if (++mmf > max_mm) break;
				}
			}
			surv = mmf <= max_mm;
		}
		unsigned long long mask = ballot(surv);
		while (mask) {
			const int sl = ctz64(mask);
			mask &= mask - 1;
			const int cq = bcast(qo0, sl), ct = bcast(to0, sl), cn = bcast(n, sl);
			int ma = 0, mm = 0;
			for (int k0 = 0; k0 < cn; k0 += 64) {
				const int k = k0 + lane;
				const bool in = k < cn;
				bool neq = false, bad = false;
				if (in) {
					neq = qseq[cq + k] != tseq[ct + k];
					if (neq) bad = !allowed(rule, qsup[cq + k], tsup[ct + k], qreads, treads);
				}
				ma += popc64(ballot(in && !neq));
				mm += popc64(ballot(bad));
				if (mm > max_mm) break;                        // :96-97
			}
			// :107 with best_ma seeded at min_overlap-1 and best_mm at max_mismatch+1, then
			// match_sort across contigs (:32-36, :239): strictly better, first wins ties
			if (mm <= max_mm && ma >= min_overlap - 1 &&
			    (!best.found || ma > best.ma || (ma == best.ma && mm < best.mm))) {
				best.found = 1; best.ma = ma; best.mm = mm; best.pos = pos; best.slot = ts;
				best.off = cq ? -cq : ct;
			}
		}
	}
}

// Corrections of one (q, t, offset), in scan order, into A.corr (contig.nim:99,:128).
// Returns the count, or -1 if it does not fit.
__device__ inline int emit_corrections(const RegionState &S, Arena &A, int qs, int ts, int off, int rule)
{
	const int lane = lane_id();
	const uint8_t *qseq = A.seq + S.off[qs], *tseq = A.seq + S.off[ts];
	const uint32_t *qsup = A.sup + S.off[qs], *tsup = A.sup + S.off[ts];
	const int qlen = S.len[qs], tlen = S.len[ts];
	const long long qreads = S.nreads[qs], treads = S.nreads[ts];
	const int qo0 = off < 0 ? -off : 0, to0 = off < 0 ? 0 : off;
	int n = qlen - qo0 < tlen - to0 ? qlen - qo0 : tlen - to0;
	int cnt = 0;
	for (int k0 = 0; k0 < n; k0 += 64) {
		const int k = k0 + lane;
		bool c = false; uint32_t a = 0, b = 0;
		if (k < n && qseq[qo0 + k] != tseq[to0 + k]) {
			a = qsup[qo0 + k]; b = tsup[to0 + k];
			c = allowed(rule, a, b, qreads, treads);
		}
		const unsigned long long m = ballot(c);
		if (c) {
			const int w = cnt + popc64(m & ((1ull << lane) - 1));
			if (w < A.corr_cap) { A.corr[w].qoff = qo0 + k; A.corr[w].toff = to0 + k; A.corr[w].qbest = a > b; }
		}
		cnt += popc64(m);
	}
	WSYNC();
	return cnt <= A.corr_cap ? cnt : -1;
}

// insert(t, q, m) of contig.nim:156-222 with the corrections in A.corr[0..ncorr).
// Mutates q as the reference does.  Returns 0, or IHP_E_CAPACITY.
__device__ inline int insert_dev(RegionState &S, Arena &A, int ts, int qs, int off, int ncorr)
{
	const int lane = lane_id();
	const int qlen = S.len[qs], tlen = S.len[ts];
	const int aoff = off < 0 ? -off : off;
	int newlen;
	if (off < 0) { newlen = aoff + tlen; if (qlen > newlen) newlen = qlen; }
	else { newlen = tlen; if (off + qlen > newlen) newlen = off + qlen; }
	if (newlen > MAXLEN) return IHP_E_CAPACITY;
	// room first: compaction moves contigs, so pointers are taken afterwards
	const bool reloc = off < 0 || newlen > S.cap[ts];
	int ncap = newlen + headroom(newlen);
	if (reloc) {
		if (!ensure_space(S, A, ncap)) {
			ncap = newlen;
			if (!ensure_space(S, A, ncap)) return IHP_E_CAPACITY;
		}
	}
	uint8_t *qseq = A.seq + S.off[qs], *tseq = A.seq + S.off[ts];
	uint32_t *qsup = A.sup + S.off[qs], *tsup = A.sup + S.off[ts];
	for (int i = lane; i < (newlen + 31) / 32; i += 64) S.bitmap[i] = 0;
	WSYNC();
	for (int c = lane; c < ncorr; c += 64) {                       // :161-173
		const Corr cr = A.corr[c];
		if (cr.qbest) { tseq[cr.toff] = qseq[cr.qoff]; tsup[cr.toff] = qsup[cr.qoff]; }
		else { qseq[cr.qoff] = tseq[cr.toff]; qsup[cr.qoff] = tsup[cr.toff]; }
		const int d = off < 0 ? cr.qoff : cr.toff;
		atomicOr(&S.bitmap[d >> 5], 1u << (d & 31));
	}
	WSYNC();
	if (off < 0) {                                                 // :180-205
		const int noff = S.bump;
		uint8_t *nseq = A.seq + noff; uint32_t *nsup = A.sup + noff;
		for (int i = lane; i < newlen; i += 64) {
			uint8_t b; uint32_t s;
			if (i < aoff) { b = qseq[i]; s = qsup[i]; }            // :184-185
			else if (i < aoff + tlen) { b = tseq[i - aoff]; s = tsup[i - aoff]; }   // :187-188
			else { b = qseq[i]; s = 0; }                           // :191-195
			if (i >= aoff && i < qlen && !((S.bitmap[i >> 5] >> (i & 31)) & 1)) s += qsup[i];   // :198-200
			nseq[i] = b; nsup[i] = s;
		}
		WSYNC();
		if (lane == 0) {
			S.off[ts] = noff; S.len[ts] = newlen; S.cap[ts] = ncap; S.bump = noff + ncap;
			S.nreads[ts] += S.nreads[qs];                          // :203
			S.start[ts] = S.start[qs];                             // :204
		}
		WSYNC();
		return 0;
	}
	if (reloc) {
		const int noff = S.bump;
		wcopy(A.seq + noff, A.sup + noff, tseq, tsup, tlen);
		if (lane == 0) { S.off[ts] = noff; S.cap[ts] = ncap; S.bump = noff + ncap; }
		WSYNC();
		tseq = A.seq + noff; tsup = A.sup + noff;
	}
	const int stop = qlen + off < newlen ? qlen + off : newlen;       // :216
	for (int i = off + lane; i < stop; i += 64) {
		const bool ext = i >= tlen;                                // i >= original_len, :220
		uint32_t s = ext ? 0u : tsup[i];                           // setLen zero-fill :212-213
		if (!((S.bitmap[i >> 5] >> (i & 31)) & 1)) {
			s += qsup[i - off];                                    // :219
			if (ext) tseq[i] = qseq[i - off];                      // :221
			tsup[i] = s;
		} else if (ext) { tseq[i] = 0; tsup[i] = 0; }
	}
	WSYNC();
	if (lane == 0) { S.len[ts] = newlen; S.nreads[ts] += S.nreads[qs]; }   // :222
	WSYNC();
	return 0;
}

// trim(c, min_support) of contig.nim:49-68: only the slot's (off,len,start) move.
__device__ inline void trim_dev(RegionState &S, const Arena &A, int s, long long min_support)
{
	const int lane = lane_id();
	const uint32_t ms = (uint32_t)min_support;
	const uint32_t *sup = A.sup + S.off[s];
	const int len = S.len[s];
	int a = len - 1 > 0 ? len - 1 : 0;                             // :52 loop exit value if nothing qualifies
	for (int b = 0; b < len - 1; b += 64) {
		const int i = b + lane;
		const unsigned long long m = ballot(i < len - 1 && sup[i] >= ms);
		if (m) { a = b + ctz64(m); break; }
	}
	if (a >= len - 1) {                                            // :56-60
		if (lane == 0) { S.start[s] += a; S.len[s] = 0; S.nreads[s] = 0; }
		WSYNC();
		return;
	}
	int bb = a;                                                    // :62-64
	for (int top = len - 1; top > a; top -= 64) {
		const int i = top - lane;
		const unsigned long long m = ballot(i > a && sup[i] >= ms);
		if (m) { bb = top - ctz64(m); break; }
	}
	if (lane == 0) {                                               // :54, :66-68
		S.start[s] += a; S.off[s] += a; S.cap[s] -= a; S.len[s] = bb - a + 1;
	}
	WSYNC();
}

// best_match (contig.nim:224-240) over `list[0..n)`.
__device__ inline Best best_match_dev(const RegionState &S, const Arena &A, int qs, const short *list, int n,
                                      int min_overlap, int max_mm)
{
	Best best = {0, 0, 0, -1, -1, 0};
	for (int i = 0; i < n; ++i) {
		const int ts = list[i];
		if (ts == qs) continue;                                    // :227
		slide_scan(S, A, qs, ts, i, min_overlap, max_mm, IHP_ALLOW_DEFAULT, best);
	}
	return best;
}

// One pass of combine (contig.nim:263-281): `in` -> `out`, returns the new count or <0.
__device__ inline int combine_pass(RegionState &S, Arena &A, short *in, int n, short *out, long long min_support,
                                   int combine_min_overlap, int max_mm)
{
	const int lane = lane_id();
	int nout = 0, usedi = 0;
	for (int i = 0; i < n; ++i) {                                  // :265-271
		const int c = in[i];
		if (min_support > 0) {
			const long long ms = S.nreads[c] < min_support ? S.nreads[c] : min_support;
			trim_dev(S, A, c, ms);
		}
		if (S.nreads[c] > 0 && nout == 0) {
			if (lane == 0) out[0] = (short)c;
			nout = 1; usedi = i;
		}
	}
	WSYNC();
	if (nout == 0) {                                               // :272
		for (int i = lane; i < n; i += 64) S.alive[in[i]] = 0;
		WSYNC();
		return 0;
	}
	for (int i = 0; i < n; ++i) {                                  // :274-281
		if (i == usedi) continue;
		const int c = in[i];
		Best b = best_match_dev(S, A, c, out, nout, combine_min_overlap, max_mm);
		if (b.found) {
			const int nc = emit_corrections(S, A, c, b.slot, b.off, IHP_ALLOW_DEFAULT);
			if (nc < 0) return IHP_E_CAPACITY;
			const int rc = insert_dev(S, A, b.slot, c, b.off, nc);
			if (rc) return rc;
			if (lane == 0) S.alive[c] = 0;
		} else if (S.nreads[c] > 0) {
			if (lane == 0) out[nout] = (short)c;
			nout++;
		} else {
			if (lane == 0) S.alive[c] = 0;
		}
		WSYNC();
	}
	return nout;
}

}  // namespace ihp
