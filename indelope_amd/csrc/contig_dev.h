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

struct Corr { int qoff; int toff; int qbest; };   // contig.nim:17

// Contig metadata of one region, in LDS, one per wave.  MC = contig slots; slot MC (QSLOT)
// describes the read currently being inserted.
template <int MC>
struct RegionStateT {
	static constexpr int MAXC = MC;
	static constexpr int QSLOT = MC;
	int off[MC + 1], len[MC + 1], cap[MC + 1];
	long long nreads[MC + 1], start[MC + 1];
	unsigned smin[MC + 1], smax[MC + 1];          // min / max support of the contig (combine phase)
	int lo3[MC + 1], hi3[MC + 1];                 // every base in [lo3, hi3) has support >= 3 (empty if lo3 >= hi3)
	unsigned char alive[MC + 1];
	short listA[MC], listB[MC];
	static constexpr int BMLEN = MC <= 64 ? 2048 : MAXLEN;   // longest contig insert_dev can merge into (longer: next pass)
	unsigned bitmap[BMLEN / 32];
	int bump;
	int err;
	long long prof[16];
};

struct Arena {
	uint8_t *seq;                // [cap] bases; LDS in the batched kernel, HBM otherwise; 4-byte aligned
	uint32_t *sup;               // [cap] supports (HBM); during the read phase: support DIFFERENCES
	int cap;                     // total elements
	int stage_off;               // [stage_off, cap) is the staging area of the read being inserted
	Corr *corr;                  // [corr_cap] corrections of the winning offset
	int corr_cap;
	long long *prof;             // optional per-wave cycle counters in LDS (diagnostics)
};

#define IHP_T0(A) const long long t0_ = (A).prof ? (long long)clock64() : 0
#define IHP_T1(A, k) do { if ((A).prof && lane_id() == 0) (A).prof[k] += (long long)clock64() - t0_; } while (0)

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

template <class ST>
__device__ __forceinline__ int alloc_slot(ST &S)
{
	const int lane = lane_id();
	for (int b = 0; b < ST::MAXC; b += 64) {
		unsigned long long m = ballot(!S.alive[b + lane]);
		if (m) return b + ctz64(m);
	}
	return -1;
}

__device__ __forceinline__ int align4(int n) { return (n + 3) & ~3; }

// Read of a difference-array entry: the entries are updated by L2 atomics, so read them at L2 too
// (a plain load could be served from a stale line of this CU's vector L1).
__device__ __forceinline__ uint32_t ld_l2(const uint32_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned ld32u(const uint32_t *b32, int byteoff)
{   // unaligned 32-bit load
	const int w = byteoff >> 2;
	return __builtin_amdgcn_alignbit(b32[w + 1], b32[w], (unsigned)(byteoff & 3) * 8u);
}

__device__ __forceinline__ void ld64u(const uint32_t *b32, int byteoff, unsigned &lo, unsigned &hi)
{
	const int w = byteoff >> 2;
	const unsigned sh = (unsigned)(byteoff & 3) * 8u;
	const unsigned w0 = b32[w], w1 = b32[w + 1], w2 = b32[w + 2];
	lo = __builtin_amdgcn_alignbit(w1, w0, sh);
	hi = __builtin_amdgcn_alignbit(w2, w1, sh);
}

template <class ST> __device__ inline bool ensure_space2(ST &S, Arena &A, int need, bool dmode);
template <class ST> __device__ inline void recompute_minmax(ST &S, const Arena &A, int s);
template <class ST> __device__ inline Best best_match_combine(const ST &S, const Arena &A, int qs, const short *list, int n,
                                                              int min_overlap, int max_mm);
constexpr int SLOT_PAD = 8;      // bytes after a slot's cap: difference entry [cap] + dword over-reads
// spare capacity of a (re)allocated contig: most contigs are single error reads that never grow, the few
// that do are relocated O(log) times
__device__ __forceinline__ int headroom(int n) { int h = n >> 1; return h < 32 ? 32 : h; }

// ---- slide_align + best_match (contig.nim:70-141, :224-240) -----------------
// Scan one target contig; update `best` under the reference's total order.
template <class ST>
__device__ inline void slide_scan(const ST &S, const Arena &A, int qs, int ts, int pos,
                                  int min_overlap, int max_mm, int rule, Best &best)
{
	const int lane = lane_id();
	const uint8_t *qseq = A.seq + S.off[qs], *tseq = A.seq + S.off[ts];
	const uint32_t *qsup = A.sup + S.off[qs], *tsup = A.sup + S.off[ts];
	const int qlen = S.len[qs], tlen = S.len[ts];
	const long long qreads = S.nreads[qs], treads = S.nreads[ts];
	const uint32_t *a32 = (const uint32_t *)A.seq;
	const int qb = S.off[qs], tb = S.off[ts];
	// (kept inside [0, 2^30]: the differences below then cannot wrap, whatever a slot holds)
	auto zone = [](int v) { return v < 0 ? 0 : v > 0x3fffffff ? 0x3fffffff : v; };
	const int qlo3 = zone(S.lo3[qs]), qhi3 = zone(S.hi3[qs]), tlo3 = zone(S.lo3[ts]), thi3 = zone(S.hi3[ts]);
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
			// A window of 8 bases where both contigs have support >= 3 everywhere: a mismatch there can
			// never be voted away (contig.nim:44-47 needs qsup < 3 or tsup < 3), so counting differing
			// bytes is exact.  Otherwise examine the first bases one by one with the supports.
			int klo = qlo3 - qo0 > tlo3 - to0 ? qlo3 - qo0 : tlo3 - to0;
			if (klo < 0) klo = 0;
			int khi = qhi3 - qo0 < thi3 - to0 ? qhi3 - qo0 : thi3 - to0;
			if (khi > n) khi = n;
			if (khi - klo >= 8) {
				unsigned a0, a1, b0, b1;
				ld64u(a32, qb + qo0 + klo, a0, a1);
				ld64u(a32, tb + to0 + klo, b0, b1);
				const unsigned x0 = a0 ^ b0, x1 = a1 ^ b1;
				const unsigned z0 = (((x0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x0) & 0x80808080u;
				const unsigned z1 = (((x1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x1) & 0x80808080u;
				surv = __popc(z0) + __popc(z1) <= max_mm;
			} else {
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
template <class ST>
__device__ inline int emit_corrections(const ST &S, Arena &A, int qs, int ts, int off, int rule)
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
template <class ST>
__device__ inline int insert_dev(ST &S, Arena &A, int ts, int qs, int off, int ncorr)
{
	const int lane = lane_id();
	const int qlen = S.len[qs], tlen = S.len[ts];
	const int aoff = off < 0 ? -off : off;
	int newlen;
	if (off < 0) { newlen = aoff + tlen; if (qlen > newlen) newlen = qlen; }
	else { newlen = tlen; if (off + qlen > newlen) newlen = off + qlen; }
	if (newlen > ST::BMLEN) return IHP_E_CAPACITY;
	// room first: compaction moves contigs, so pointers are taken afterwards
	const bool reloc = off < 0 || newlen > S.cap[ts];
	int ncap = align4(newlen + headroom(newlen));
	if (reloc) {
		if (!ensure_space2(S, A, ncap, false)) {
			ncap = align4(newlen);
			if (!ensure_space2(S, A, ncap, false)) return IHP_E_CAPACITY;
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
			S.off[ts] = noff; S.len[ts] = newlen; S.cap[ts] = ncap; S.bump = noff + ncap + SLOT_PAD;
			S.nreads[ts] += S.nreads[qs];                          // :203
			S.start[ts] = S.start[qs];                             // :204
		}
		WSYNC();
		return 0;
	}
	if (reloc) {
		const int noff = S.bump;
		wcopy(A.seq + noff, A.sup + noff, tseq, tsup, tlen);
		if (lane == 0) { S.off[ts] = noff; S.cap[ts] = ncap; S.bump = noff + ncap + SLOT_PAD; }
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
template <class ST>
__device__ inline void trim_dev(ST &S, const Arena &A, int s, long long min_support)
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
template <class ST>
__device__ inline Best best_match_dev(const ST &S, const Arena &A, int qs, const short *list, int n,
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
template <class ST>
__device__ inline int combine_pass(ST &S, Arena &A, short *in, int n, short *out, long long min_support,
                                   int combine_min_overlap, int max_mm)
{
	const int lane = lane_id();
	int nout = 0, usedi = 0;
	for (int i = 0; i < n; ++i) {                                  // :265-271
		const int c = in[i];
		if (min_support > 0) {
			const long long ms = S.nreads[c] < min_support ? S.nreads[c] : min_support;
			IHP_T0(A);
			trim_dev(S, A, c, ms);
			recompute_minmax(S, A, c);
			IHP_T1(A, 7);
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
		Best b = best_match_combine(S, A, c, out, nout, combine_min_overlap, max_mm);
		if (b.found) {
			IHP_T0(A);
			const int nc = emit_corrections(S, A, c, b.slot, b.off, IHP_ALLOW_DEFAULT);
			if (nc < 0) return IHP_E_CAPACITY;
			const int rc = insert_dev(S, A, b.slot, c, b.off, nc);
			if (rc) return rc;
			if (lane == 0) S.alive[c] = 0;
			recompute_minmax(S, A, b.slot);
			IHP_T1(A, 6);
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


// ============================================================================================
// Fast paths.  (1) Exact-match scan: with max_mismatch == 0 and no position where the vote rule
// (contig.nim:44-47) can fire, slide_align reduces to "longest exactly matching overlap, first in
// scan order", done here on 8 / 4 bases per lane with dword loads from the 4-byte aligned arena.
// (2) Read phase of assemble (indelope.nim:163-169): every query is a fresh read (support 1
// everywhere, nreads 1), for which the vote rule can never fire because every contig base has
// support >= 1; `t.support[i] += q.support[j]` over the overlap is then a +1 on a range, kept as a
// DIFFERENCE array in A.sup (two scalar updates per insert) and turned into supports by one prefix
// sum per contig before combine (materialize_supports).
// ============================================================================================

// Can allowable_mismatch (default rule) be true for any pair of positions of q and t?  Necessary
// condition from the per-contig support extrema; false => the exact scan is equivalent.
template <class ST>
__device__ __forceinline__ bool may_allow(const ST &S, int qs, int ts)
{
	const unsigned qmin = S.smin[qs], qmax = S.smax[qs], tmin = S.smin[ts], tmax = S.smax[ts];
	const long long qreads = S.nreads[qs], treads = S.nreads[ts];
	const bool p1 = qmin < 3u && tmax > 3u * qmin && qreads > 3ll * (long long)qmin;
	const bool p2 = tmin < 3u && qmax > 3u * tmin && treads > 3ll * (long long)tmin;
	return p1 || p2;
}

// slide_align for max_mismatch == 0 without votes (contig.nim:70-141), same total order as slide_scan:
// one offset per lane, any lengths (the general form; slide_scan_exact below is the usual one).
template <class ST>
__device__ inline void slide_scan_exact_small(const ST &S, const Arena &A, int qs, int ts, int pos, int min_overlap, Best &best)
{
	const int lane = lane_id();
	const uint32_t *a32 = (const uint32_t *)A.seq;
	const int qb = S.off[qs], tb = S.off[ts];
	const int qlen = S.len[qs], tlen = S.len[ts];
	const int omax = tlen - min_overlap;
	int omin_abs = qlen - min_overlap;
	if (omin_abs < 0) omin_abs = -omin_abs;
	const int n1 = omax >= 0 ? omax + 1 : 0;
	const int total = n1 + omin_abs;
	if (total <= 0) return;
	unsigned q0, q1, t0, t1;                         // first 8 bases of q and of t (wave-uniform)
	ld64u(a32, qb, q0, q1);
	ld64u(a32, tb, t0, t1);
	for (int base = 0; base < total; base += 64) {
		const int idx = base + lane;
		const bool ph1 = idx < n1;
		const int o = ph1 ? idx : idx - n1 + 1;
		int n = ph1 ? (qlen < tlen - o ? qlen : tlen - o) : (qlen - o < tlen ? qlen - o : tlen);
		if (n < 0) n = 0;
		// accepted offsets have mm == 0, so a later one only wins with strictly more matches (:107)
		const int need = best.found ? best.ma + 1 : min_overlap - 1;
		bool surv = idx < total && n >= need;
		if (surv) {
			unsigned lo, hi;
			ld64u(a32, ph1 ? tb + o : qb + o, lo, hi);
			unsigned x0 = lo ^ (ph1 ? q0 : t0), x1 = hi ^ (ph1 ? q1 : t1);
			if (n < 8) {
				if (n <= 4) { x1 = 0; x0 = n == 4 ? x0 : (n == 0 ? 0u : x0 & ((1u << (8 * n)) - 1u)); }
				else x1 &= (1u << (8 * (n - 4))) - 1u;
			}
			surv = (x0 | x1) == 0;
		}
		unsigned long long mask = ballot(surv);
		while (mask) {
			const int sl = ctz64(mask);
			mask &= mask - 1;
			const int co = bcast(o, sl), cn = bcast(n, sl), cph = bcast((int)ph1, sl);
			if (best.found && cn <= best.ma) continue;          // uniform
			const int cq = cph ? 0 : co, ct = cph ? co : 0;
			bool ok = true;
			for (int k0 = 0; k0 < cn; k0 += 256) {
				const int k = k0 + 4 * lane;
				bool bad = false;
				if (k < cn) {
					unsigned x = ld32u(a32, qb + cq + k) ^ ld32u(a32, tb + ct + k);
					const int rem = cn - k;
					if (rem < 4) x &= (1u << (8 * rem)) - 1u;
					bad = x != 0;
				}
				if (ballot(bad)) { ok = false; break; }
			}
			if (ok && cn >= min_overlap - 1 && (!best.found || cn > best.ma)) {
				best.found = 1; best.ma = cn; best.mm = 0; best.pos = pos; best.slot = ts;
				best.off = cq ? -cq : ct;
			}
		}
	}
}

// ---- exact scan, four offsets per lane ------------------------------------------------------
// Everything about the two contigs is passed in wave-uniform (scalar) form: arena byte offsets qb/tb, lengths,
// the first 8 bases of each.  An offset phase looks at offsets o in [o_lo, o_hi] of a `stream` (byte offset sb,
// slen bases) against the first bases of the `other` contig (ob, olen); its overlap n(o) = min(olen, slen - o) is
// >= 8 for every o in range.  Each lane filters FOUR consecutive offsets from one 16-byte read of the stream
// (funnel shifts give the four 8-byte windows), so a 150-base read against a 250-base contig is one LDS round trip
// per phase instead of five.  Survivors are verified in ascending offset order (first-wins tie rule of
// contig.nim:107).
struct ScanSide { int b, len; unsigned h0, h1; };            // arena byte offset, bases, first 8 bases

// window filter of one 256-offset chunk: bit j of the result <-> offset o_lo + base + 4*lane + j matches on 8 bases
__device__ __forceinline__ unsigned exact_hits(const uint32_t *a32, int sb, int o_lo, int o_hi, int base, unsigned p0, unsigned p1)
{
	const int lane = lane_id();
	const int a0 = sb + o_lo;
	const unsigned sh = (unsigned)(a0 & 3) * 8u;
	const int ofs = base + 4 * lane;                         // this lane's first offset, relative to o_lo
	unsigned hits = 0;
	if (o_lo + ofs <= o_hi) {
		const int d = (a0 >> 2) + (ofs >> 2);
		const unsigned w0 = a32[d], w1 = a32[d + 1], w2 = a32[d + 2], w3 = a32[d + 3];
		const unsigned u0 = __builtin_amdgcn_alignbit(w1, w0, sh), u1 = __builtin_amdgcn_alignbit(w2, w1, sh);
		const unsigned u2 = __builtin_amdgcn_alignbit(w3, w2, sh);
		const unsigned x0 = (u0 ^ p0) | (u1 ^ p1);
		const unsigned x1 = (__builtin_amdgcn_alignbit(u1, u0, 8) ^ p0) | (__builtin_amdgcn_alignbit(u2, u1, 8) ^ p1);
		const unsigned x2 = (__builtin_amdgcn_alignbit(u1, u0, 16) ^ p0) | (__builtin_amdgcn_alignbit(u2, u1, 16) ^ p1);
		const unsigned x3 = (__builtin_amdgcn_alignbit(u1, u0, 24) ^ p0) | (__builtin_amdgcn_alignbit(u2, u1, 24) ^ p1);
		if (min(min(x0, x1), min(x2, x3)) == 0) {            // rare: form the per-offset bits (offsets past o_hi excluded)
			const int left = o_hi - (o_lo + ofs);            // >= 0
			hits = (x0 == 0 ? 1u : 0u) | (x1 == 0 && left >= 1 ? 2u : 0u) | (x2 == 0 && left >= 2 ? 4u : 0u) | (x3 == 0 && left >= 3 ? 8u : 0u);
		}
	}
	return hits;
}

// verify the survivors of one chunk in ascending offset order and keep the best (contig.nim:103-111, :128-135)
__device__ __forceinline__ void exact_verify(const uint32_t *a32, unsigned hits, int sb, int slen, int ob, int olen, int o_base,
                                              bool neg, int min_overlap, int pos, int ts, Best &best)
{
	const int lane = lane_id();
	unsigned long long mask = ballot(hits != 0);
	while (mask) {                                           // lanes ascending, then j ascending: ascending offsets
		const int sl = ctz64(mask);
		mask &= mask - 1;
		unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)hits, sl);
		while (bits) {
			const int j = __builtin_ctz(bits);
			bits &= bits - 1;
			const int co = o_base + 4 * sl + j;
			int cn = slen - co; cn = cn < olen ? cn : olen;
			if (best.found && cn <= best.ma) continue;       // accepted offsets have mm == 0: only strictly more matches win (:107)
			bool ok = true;
			for (int k0 = 0; k0 < cn; k0 += 256) {
				const int k = k0 + 4 * lane;
				bool bad = false;
				if (k < cn) {
					unsigned x = ld32u(a32, ob + k) ^ ld32u(a32, sb + co + k);
					const int rem = cn - k;
					if (rem < 4) x &= (1u << (8 * rem)) - 1u;
					bad = x != 0;
				}
				if (ballot(bad)) { ok = false; break; }
			}
			if (ok && cn >= min_overlap - 1) {
				best.found = 1; best.ma = cn; best.mm = 0; best.pos = pos; best.slot = ts;
				best.off = neg ? -co : co;                   // the offset is on the query (contig.nim:114-135): reported as -o
			}
		}
	}
}

// slide_align for max_mismatch == 0 without votes (contig.nim:70-141), same total order as slide_scan.
// Preconditions (else slide_scan_exact_small): min_overlap >= 9, q.len >= min_overlap, t.len >= 8.
__device__ __forceinline__ void slide_scan_exact_u(const Arena &A, const ScanSide q, const ScanSide t, int pos, int ts, int min_overlap, Best &best)
{
	const uint32_t *a32 = (const uint32_t *)A.seq;
	const int omax = t.len - min_overlap;                    // :79: offsets 0..omax on the target (:81-111)
	const int omin = q.len - min_overlap;                    // :78: then offsets 1..omin on the query (:114-135)
	if (omax < 256 && omin <= 256) {                         // the usual sizes: both filters issued back to back
		const unsigned h1 = omax >= 0 ? exact_hits(a32, t.b, 0, omax, 0, q.h0, q.h1) : 0u;
		const unsigned h2 = omin >= 1 ? exact_hits(a32, q.b, 1, omin, 0, t.h0, t.h1) : 0u;
		exact_verify(a32, h1, t.b, t.len, q.b, q.len, 0, false, min_overlap, pos, ts, best);
		exact_verify(a32, h2, q.b, q.len, t.b, t.len, 1, true, min_overlap, pos, ts, best);
		return;
	}
	for (int base = 0; base <= omax; base += 256)
		exact_verify(a32, exact_hits(a32, t.b, 0, omax, base, q.h0, q.h1), t.b, t.len, q.b, q.len, base, false, min_overlap, pos, ts, best);
	for (int base = 0; 1 + base <= omin; base += 256)
		exact_verify(a32, exact_hits(a32, q.b, 1, omin, base, t.h0, t.h1), q.b, q.len, t.b, t.len, 1 + base, true, min_overlap, pos, ts, best);
}

__device__ __forceinline__ ScanSide scan_side_uni(const uint32_t *a32, int b, int len)
{
	ScanSide s;
	s.b = uni(b); s.len = uni(len);
	unsigned h0, h1;
	ld64u(a32, s.b, h0, h1);
	s.h0 = (unsigned)uni((int)h0); s.h1 = (unsigned)uni((int)h1);
	return s;
}

template <class ST>
__device__ inline void slide_scan_exact(const ST &S, const Arena &A, int qs, int ts, int pos, int min_overlap, Best &best)
{
	const int qlen = uni(S.len[qs]), tlen = uni(S.len[ts]);
	min_overlap = uni(min_overlap);
	if (min_overlap < 9 || qlen < min_overlap || tlen < 8) {  // windows shorter than 8 bases, or the abs(omin) oddity of :114
		slide_scan_exact_small(S, A, qs, ts, pos, min_overlap, best);
		return;
	}
	const uint32_t *a32 = (const uint32_t *)A.seq;
	slide_scan_exact_u(A, scan_side_uni(a32, S.off[qs], qlen), scan_side_uni(a32, S.off[ts], tlen), uni(pos), uni(ts), min_overlap, best);
}

// Support extrema and the clean zone [lo3, hi3) (every support >= 3) of one contig.
template <class ST>
__device__ inline void recompute_minmax(ST &S, const Arena &A, int s)
{
	const int lane = lane_id();
	const uint32_t *sup = A.sup + S.off[s];
	const int n = S.len[s];
	unsigned mn = 0xffffffffu, mx = 0;
	int first = 0x7fffffff, last = -1, cnt = 0;
	for (int i = lane; i < n; i += 64) {
		const unsigned v = sup[i];
		mn = v < mn ? v : mn; mx = v > mx ? v : mx;
		if (v >= 3u) { first = i < first ? i : first; last = i > last ? i : last; cnt++; }
	}
	mn = wave_min_u32(mn); mx = wave_max_u32(mx);
	first = wave_min_i32(first); last = wave_max_i32s(last); cnt = wave_sum_i(cnt);
	if (lane == 0) {
		S.smin[s] = mn; S.smax[s] = mx;
		const bool clean = last >= first && cnt == last - first + 1;    // no dip below 3 inside
		// sentinel small enough that lo3 - offset never overflows
		S.lo3[s] = clean ? first : 0x3fffffff; S.hi3[s] = clean ? last + 1 : 0;
	}
	WSYNC();
}

// Difference arrays -> supports, in place, plus the support extrema and clean zone of recompute_minmax()
// from the values while they are in registers (end of the read phase).
template <class ST>
__device__ inline void materialize_supports(ST &S, Arena &A, const short *list, int n)
{
	const int lane = lane_id();
	n = uni(n);
	for (int c = 0; c < n; ++c) {
		const int s = uni((int)list[c]);
		uint32_t *d = A.sup + uni(S.off[s]);
		const int len = uni(S.len[s]);
		unsigned carry = 0, mn = 0xffffffffu, mx = 0;
		int first = 0x7fffffff, last = -1, cnt = 0;
		for (int i0 = 0; i0 < len; i0 += 64) {
			const int i = i0 + lane;
			unsigned v = i < len ? ld_l2(&d[i]) : 0u;
			v = wave_scan_add(v) + carry;
			if (i < len) {
				d[i] = v; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
				if (v >= 3u) { first = i < first ? i : first; last = i; cnt++; }
			}
			carry = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
		}
		mn = wave_min_u32(mn); mx = wave_max_u32(mx);
		first = wave_min_i32(first); last = wave_max_i32s(last); cnt = wave_sum_i(cnt);
		if (lane == 0) {
			S.smin[s] = mn; S.smax[s] = mx;
			const bool clean = last >= first && cnt == last - first + 1;    // no dip below 3 inside
			S.lo3[s] = clean ? first : 0x3fffffff; S.hi3[s] = clean ? last + 1 : 0;
		}
	}
	WSYNC();
}

// Close the holes of the arena (read-phase flavour keeps the difference arrays valid: len+1 entries
// move, the rest of the new capacity is zeroed).
template <class ST>
__device__ inline void compact2(ST &S, Arena &A, bool dmode)
{
	const int lane = lane_id();
	int newbump = 0;
	long long last = -1;
	for (;;) {
		long long key = 0x7fffffffffffffffll;
		for (int b = 0; b < ST::MAXC; b += 64) {
			const int s = b + lane;
			if (S.alive[s] && (long long)S.off[s] > last) {
				const long long k = ((long long)S.off[s] << 16) | s;
				key = k < key ? k : key;
			}
		}
		key = wave_min_ll(key);
		if (key == 0x7fffffffffffffffll) break;
		const int s = (int)(key & 0xffff), o = (int)(key >> 16), n = S.len[s];
		const int ncap = align4(n);
		WSYNC();
		if (o != newbump) {
			// ascending 64-element chunks; dst < src so a chunk is fully read before it is overwritten
			const int nd = dmode ? n + 1 : n;
			for (int i0 = 0; i0 < nd; i0 += 64) {
				const int i = i0 + lane;
				uint8_t b = 0; uint32_t v = 0;
				if (i < n) b = A.seq[o + i];
				if (i < nd) v = dmode ? ld_l2(&A.sup[o + i]) : A.sup[o + i];
				WSYNC();
				if (i < n) A.seq[newbump + i] = b;
				if (i < nd) A.sup[newbump + i] = v;
			}
			WSYNC();
		}
		if (dmode) for (int i = n + 1 + lane; i <= ncap; i += 64) A.sup[newbump + i] = 0;
		if (lane == 0) { S.off[s] = newbump; S.cap[s] = ncap; }
		WSYNC();
		newbump += ncap + SLOT_PAD;
		last = o;
	}
	if (lane == 0) S.bump = newbump;
	WSYNC();
}

template <class ST>
__device__ inline bool ensure_space2(ST &S, Arena &A, int need, bool dmode)
{
	if (S.bump + need + SLOT_PAD <= A.stage_off) return true;
	compact2(S, A, dmode);
	return S.bump + need + SLOT_PAD <= A.stage_off;
}

// New contig from the staged read (contig.nim:143-150, :248), read-phase representation.
template <class ST>
__device__ inline int new_contig_from_read(ST &S, Arena &A, int &slot_out)
{
	const int lane = lane_id();
	const int tl = S.len[ST::QSLOT];
	const int slot = alloc_slot(S);
	if (slot < 0) return IHP_E_CAPACITY;
	int cap = align4(tl + headroom(tl));
	if (!ensure_space2(S, A, cap, true)) { cap = align4(tl); if (!ensure_space2(S, A, cap, true)) return IHP_E_CAPACITY; }
	const int noff = S.bump;
	for (int i = lane; i < tl; i += 64) A.seq[noff + i] = A.seq[A.stage_off + i];
	for (int i = lane; i <= cap; i += 64) A.sup[noff + i] = i == 0 ? 1u : (i == tl ? 0xffffffffu : 0u);   // +1 on [0, tl)
	if (tl == 0 && lane == 0) A.sup[noff] = 0;
	if (lane == 0) {
		S.off[slot] = noff; S.len[slot] = tl; S.cap[slot] = cap; S.nreads[slot] = 1;
		S.start[slot] = S.start[ST::QSLOT]; S.alive[slot] = 1; S.bump = noff + cap + SLOT_PAD;
	}
	WSYNC();
	slot_out = slot;
	return 0;
}

// insert(t, q, m) of contig.nim:156-222 for a fresh read q (no corrections possible), with the
// supports kept as differences.
template <class ST>
__device__ inline int insert_read(ST &S, Arena &A, int ts, int off)
{
	const int lane = lane_id();
	const int qs = ST::QSLOT;
	const int qlen = S.len[qs], tlen = S.len[ts];
	const int aoff = off < 0 ? -off : off;
	int newlen;
	if (off < 0) { newlen = aoff + tlen; if (qlen > newlen) newlen = qlen; }
	else { newlen = tlen; if (off + qlen > newlen) newlen = off + qlen; }
	if (newlen > MAXLEN) return IHP_E_CAPACITY;
	const bool reloc = off < 0 || newlen > S.cap[ts];
	int ncap = align4(newlen + headroom(newlen));
	if (reloc) {
		if (!ensure_space2(S, A, ncap, true)) { ncap = align4(newlen); if (!ensure_space2(S, A, ncap, true)) return IHP_E_CAPACITY; }
	}
	const uint8_t *qseq = A.seq + S.off[qs];
	uint8_t *tseq = A.seq + S.off[ts];
	uint32_t *td = A.sup + S.off[ts];
	if (reloc) WSYNC();                                            // earlier range updates must have landed
	if (off < 0) {                                                 // :180-205
		const int noff = S.bump;
		uint8_t *nseq = A.seq + noff; uint32_t *nd = A.sup + noff;
		for (int i = lane; i < newlen; i += 64)
			nseq[i] = i < aoff ? qseq[i] : (i < aoff + tlen ? tseq[i - aoff] : qseq[i]);
		for (int i = lane; i <= ncap; i += 64) {
			uint32_t v = (i >= aoff && i - aoff <= tlen) ? ld_l2(&td[i - aoff]) : 0u;
			if (i == 0) v += 1u;                                   // q covers [0, qlen) of the new contig
			if (i == qlen) v -= 1u;
			nd[i] = v;
		}
		WSYNC();
		if (lane == 0) {
			S.off[ts] = noff; S.len[ts] = newlen; S.cap[ts] = ncap; S.bump = noff + ncap + SLOT_PAD;
			S.nreads[ts] += 1;                                     // :203
			S.start[ts] = S.start[qs];                             // :204
		}
		WSYNC();
		return 0;
	}
	if (reloc) {
		const int noff = S.bump;
		uint8_t *nseq = A.seq + noff; uint32_t *nd = A.sup + noff;
		for (int i = lane; i < tlen; i += 64) nseq[i] = tseq[i];
		for (int i = lane; i <= ncap; i += 64) nd[i] = i <= tlen ? ld_l2(&td[i]) : 0u;
		WSYNC();
		if (lane == 0) { S.off[ts] = noff; S.cap[ts] = ncap; S.bump = noff + ncap + SLOT_PAD; }
		WSYNC();
		tseq = nseq; td = nd;
	}
	for (int i = tlen + lane; i < newlen; i += 64) tseq[i] = qseq[i - off];   // :220-221
	if (lane == 0) {
		// :216-219 as a range update; result-less atomics so nothing waits on HBM (the arrays are only
		// read after a WSYNC: relocation, compaction, materialize_supports)
		__hip_atomic_fetch_add(&td[off], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_fetch_add(&td[off + qlen], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		S.len[ts] = newlen; S.nreads[ts] += 1;                     // :222
	}
	LDS_ORDER();
	return 0;
}

// best_match (contig.nim:224-240) for a fresh read against list[0..n), one contig after the other: the slot
// metadata and first 8 bases of up to 64 contigs are gathered lane-parallel and handed to the scan through
// v_readlane.  The general form (any lengths); best_match_read below is the usual one.
template <class ST>
__device__ inline Best best_match_read_seq(const ST &S, const Arena &A, const short *list, int n, int min_overlap)
{
	const int lane = lane_id();
	Best best = {0, 0, 0, -1, -1, 0};
	n = uni(n); min_overlap = uni(min_overlap);
	const int qlen = uni(S.len[ST::QSLOT]);
	if (min_overlap < 9 || qlen < min_overlap) {
		for (int i = 0; i < n; ++i) slide_scan_exact_small(S, A, ST::QSLOT, list[i], i, min_overlap, best);
		return best;
	}
	const uint32_t *a32 = (const uint32_t *)A.seq;
	const ScanSide q = scan_side_uni(a32, S.off[ST::QSLOT], qlen);
	for (int c0 = 0; c0 < n; c0 += 64) {
		int m_ts = 0, m_off = 0, m_len = 0; unsigned m_h0 = 0, m_h1 = 0;
		if (c0 + lane < n) {
			m_ts = list[c0 + lane]; m_off = S.off[m_ts]; m_len = S.len[m_ts];
			ld64u(a32, m_off, m_h0, m_h1);
		}
		const int m = n - c0 < 64 ? n - c0 : 64;
		for (int i = 0; i < m; ++i) {
			ScanSide t;
			t.b = __builtin_amdgcn_readlane(m_off, i); t.len = __builtin_amdgcn_readlane(m_len, i);
			t.h0 = (unsigned)__builtin_amdgcn_readlane((int)m_h0, i); t.h1 = (unsigned)__builtin_amdgcn_readlane((int)m_h1, i);
			const int ts = __builtin_amdgcn_readlane(m_ts, i);
			if (t.len < 8) { slide_scan_exact_small(S, A, ST::QSLOT, ts, c0 + i, min_overlap, best); continue; }
			slide_scan_exact_u(A, q, t, c0 + i, ts, min_overlap, best);
		}
	}
	return best;
}

// Candidate (cn bases of the arena at xb and yb equal?) against the best so far under the reference's total
// order: more matches win (contig.nim:107), ties go to the earlier contig (match_sort :32-36, :239), within a
// contig to the target-offset phase (:81-111) before the query-offset phase (:114-135), then the smaller offset.
struct BestOrd { Best b; int ph, o; };
__device__ __forceinline__ void consider(const uint32_t *a32, BestOrd &B, int cn, int pos, int ph, int o, int slot, int xb, int yb, int min_overlap)
{
	const int lane = lane_id();
	if (cn < min_overlap - 1) return;
	if (B.b.found) {
		if (cn < B.b.ma) return;
		if (cn == B.b.ma && (pos > B.b.pos || (pos == B.b.pos && (ph > B.ph || (ph == B.ph && o >= B.o))))) return;
	}
	for (int k0 = 0; k0 < cn; k0 += 256) {
		const int k = k0 + 4 * lane;
		bool bad = false;
		if (k < cn) {
			unsigned x = ld32u(a32, xb + k) ^ ld32u(a32, yb + k);
			const int rem = cn - k;
			if (rem < 4) x &= (1u << (8 * rem)) - 1u;
			bad = x != 0;
		}
		if (ballot(bad)) return;
	}
	B.b.found = 1; B.b.ma = cn; B.b.mm = 0; B.b.pos = pos; B.b.slot = slot; B.b.off = ph ? -o : o;
	B.ph = ph; B.o = o;
}

// best_match for the combine phase, one contig after the other: exact scan when the vote rule cannot fire for
// the pair, else the generic scan.  The general form; best_match_combine below is the usual one.
template <class ST>
__device__ inline Best best_match_combine_seq(const ST &S, const Arena &A, int qs, const short *list, int n,
                                          int min_overlap, int max_mm)
{
	Best best = {0, 0, 0, -1, -1, 0};
	for (int i = 0; i < n; ++i) {
		const int ts = list[i];
		if (ts == qs) continue;                                    // :227
		if (max_mm == 0 && !may_allow(S, qs, ts)) { IHP_T0(A); slide_scan_exact(S, A, qs, ts, i, min_overlap, best); IHP_T1(A, 5); }
		else { IHP_T0(A); slide_scan(S, A, qs, ts, i, min_overlap, max_mm, IHP_ALLOW_DEFAULT, best); IHP_T1(A, 4); }
	}
	return best;
}

// best_match (contig.nim:224-240) against all contigs of the list at once, for queries the exact scan is valid
// for: a fresh read (COMBINE = false) or, in the combine phase, a contig for which may_allow() is false against
// every contig of the list (anything else goes to the one-after-the-other forms).
//   Target-offset phase (:81-111): the valid offsets [0, len - min_overlap] of the contigs are laid end to end in
//   groups of four; a lane takes one group (its contig found by walking the scalar prefix sums), reads 16 bytes of
//   that contig and filters four 8-base windows against the head of the query.
//   Query-offset phase (:114-135): lane o holds the query's 8 bases from offset o (64 offsets at a time) and the
//   contigs' first 8 bases are broadcast one after the other -- or, when there are fewer offsets than contigs, one
//   lane per contig and the windows are broadcast.
// Survivors are verified on the full overlap and ranked by consider().
template <class ST, bool COMBINE>
__device__ inline Best best_match_all(const ST &S, const Arena &A, int qs, const short *list, int n, int min_overlap, int max_mm)
{
	const int lane = lane_id();
	n = uni(n); min_overlap = uni(min_overlap); qs = uni(qs);
	const int qlen = uni(S.len[qs]);
	const int omin = qlen - min_overlap;                     // :78
	if (min_overlap < 9 || omin < 0 || (COMBINE && max_mm != 0))
		return COMBINE ? best_match_combine_seq(S, A, qs, list, n, min_overlap, max_mm) : best_match_read_seq(S, A, list, n, min_overlap);
	const uint32_t *a32 = (const uint32_t *)A.seq;
	const int qb = uni(S.off[qs]);
	unsigned rw0, rw1;                                       // lane o: the query's 8 bases from offset o
	ld64u(a32, qb + (lane <= omin ? lane : 0), rw0, rw1);
	const unsigned qh0 = (unsigned)__builtin_amdgcn_readlane((int)rw0, 0), qh1 = (unsigned)__builtin_amdgcn_readlane((int)rw1, 0);
	unsigned qmin = 0, qmax = 0; long long qreads = 0;
	if (COMBINE) { qmin = (unsigned)uni((int)S.smin[qs]); qmax = (unsigned)uni((int)S.smax[qs]); qreads = uni(S.nreads[qs]); }
	BestOrd B; B.b = {0, 0, 0, -1, -1, 0}; B.ph = 0; B.o = 0;
	Best G = {0, 0, 0, -1, -1, 0};                           // best of the pairs that need the generic scan
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int m = n - c0 < 64 ? n - c0 : 64;
		int m_ts = 0, m_off = 0, m_len = 0; unsigned m_h0 = 0, m_h1 = 0;
		bool use = lane < m;
		if (use) {
			m_ts = list[c0 + lane]; m_off = S.off[m_ts]; m_len = S.len[m_ts];
			ld64u(a32, m_off, m_h0, m_h1);
		}
		if (ballot(use && m_len < 8))                        // windows shorter than 8 bases
			return COMBINE ? best_match_combine_seq(S, A, qs, list, n, min_overlap, max_mm) : best_match_read_seq(S, A, list, n, min_overlap);
		if (COMBINE) {
			bool votes = false;                              // may_allow(): can the vote rule fire for this pair at all?
			if (use) {
				if (m_ts == qs) use = false;                 // :227
				else {
					const unsigned tmin = S.smin[m_ts], tmax = S.smax[m_ts];
					const long long treads = S.nreads[m_ts];
					votes = (qmin < 3u && tmax > 3u * qmin && qreads > 3ll * (long long)qmin) ||
					        (tmin < 3u && qmax > 3u * tmin && treads > 3ll * (long long)tmin);
					if (votes) use = false;
				}
			}
			// those pairs get the generic scan, one after the other; its best joins the ranking at the end
			unsigned long long gm = ballot(votes);
			while (gm) {
				const int i = ctz64(gm);
				gm &= gm - 1;
				IHP_T0(A);
				slide_scan(S, A, qs, __builtin_amdgcn_readlane(m_ts, i), c0 + i, min_overlap, max_mm, IHP_ALLOW_DEFAULT, G);
				IHP_T1(A, 4);
			}
		}
		// ---- offsets on the contigs
		const int n1 = m_len - min_overlap + 1;              // offsets 0 .. len - min_overlap (:79)
		const unsigned nq = use && n1 > 0 ? (unsigned)(n1 + 3) >> 2 : 0u;
		const unsigned incl = wave_scan_add(nq), excl = incl - nq;
		const int Q = __builtin_amdgcn_readlane((int)incl, 63);
		int i0 = 0;                                          // first contig whose groups are not all behind us
		for (int g0 = 0; g0 < Q; g0 += 64) {
			const int g = g0 + lane;
			int c_i = 0;
			for (int i = i0; i < m; ++i) {                   // contigs in list order have ascending group ranges
				const int ei = __builtin_amdgcn_readlane((int)excl, i);
				if (ei >= g0 + 64) break;
				c_i = g >= ei ? i : c_i;                     // the last contig that starts at or before g owns it
				if (__builtin_amdgcn_readlane((int)incl, i) <= g0 + 64) i0 = i + 1;
			}
			// (contigs with no valid offset have empty ranges: a later contig with the same start overrides them)
			const int c_off = __builtin_amdgcn_ds_bpermute(c_i << 2, m_off), c_n1 = __builtin_amdgcn_ds_bpermute(c_i << 2, n1);
			const int c_ex = __builtin_amdgcn_ds_bpermute(c_i << 2, (int)excl);
			const int o1 = 4 * (g - c_ex);                   // first of this lane's four offsets
			unsigned hits = 0;
			if (g < Q && o1 < c_n1) {
				const int ab = c_off + o1;
				const unsigned sh = (unsigned)(ab & 3) * 8u;
				const int d = ab >> 2;
				const unsigned w0 = a32[d], w1 = a32[d + 1], w2 = a32[d + 2], w3 = a32[d + 3];
				const unsigned u0 = __builtin_amdgcn_alignbit(w1, w0, sh), u1 = __builtin_amdgcn_alignbit(w2, w1, sh);
				const unsigned u2 = __builtin_amdgcn_alignbit(w3, w2, sh);
				const unsigned x0 = (u0 ^ qh0) | (u1 ^ qh1);
				const unsigned x1 = (__builtin_amdgcn_alignbit(u1, u0, 8) ^ qh0) | (__builtin_amdgcn_alignbit(u2, u1, 8) ^ qh1);
				const unsigned x2 = (__builtin_amdgcn_alignbit(u1, u0, 16) ^ qh0) | (__builtin_amdgcn_alignbit(u2, u1, 16) ^ qh1);
				const unsigned x3 = (__builtin_amdgcn_alignbit(u1, u0, 24) ^ qh0) | (__builtin_amdgcn_alignbit(u2, u1, 24) ^ qh1);
				if (min(min(x0, x1), min(x2, x3)) == 0) {
					const int left = c_n1 - 1 - o1;          // >= 0: offsets past len - min_overlap are not candidates
					hits = (x0 == 0 ? 1u : 0u) | (x1 == 0 && left >= 1 ? 2u : 0u) | (x2 == 0 && left >= 2 ? 4u : 0u) | (x3 == 0 && left >= 3 ? 8u : 0u);
				}
			}
			unsigned long long mask = ballot(hits != 0);
			while (mask) {
				const int sl = ctz64(mask);
				mask &= mask - 1;
				unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)hits, sl);
				const int i = __builtin_amdgcn_readlane(c_i, sl), ob = __builtin_amdgcn_readlane(o1, sl);
				const int tb = __builtin_amdgcn_readlane(m_off, i), tlen = __builtin_amdgcn_readlane(m_len, i);
				const int ts = __builtin_amdgcn_readlane(m_ts, i);
				while (bits) {
					const int o = ob + __builtin_ctz(bits);
					bits &= bits - 1;
					const int cn = qlen < tlen - o ? qlen : tlen - o;
					consider(a32, B, cn, c0 + i, 0, o, ts, qb, tb + o, min_overlap);
				}
			}
		}
		// ---- offsets on the query: lanes are contigs and the windows are broadcast, or the other way round
		if (omin < m && omin <= 63) {
			for (int o = 1; o <= omin; ++o) {
				const unsigned s0 = (unsigned)__builtin_amdgcn_readlane((int)rw0, o), s1 = (unsigned)__builtin_amdgcn_readlane((int)rw1, o);
				unsigned long long mask = ballot(use && ((m_h0 ^ s0) | (m_h1 ^ s1)) == 0);
				while (mask) {
					const int i = ctz64(mask);
					mask &= mask - 1;
					const int tb = __builtin_amdgcn_readlane(m_off, i), tlen = __builtin_amdgcn_readlane(m_len, i);
					const int cn = qlen - o < tlen ? qlen - o : tlen;
					consider(a32, B, cn, c0 + i, 1, o, __builtin_amdgcn_readlane(m_ts, i), qb + o, tb, min_overlap);
				}
			}
		} else {
			const unsigned long long usem = ballot(use);
			for (int ob = 0; ob <= omin; ob += 64) {         // lane <-> offset ob + lane
				unsigned w0 = rw0, w1 = rw1;
				if (ob) ld64u(a32, qb + (ob + lane <= omin ? ob + lane : 0), w0, w1);
				const bool valid = ob + lane >= 1 && ob + lane <= omin;
				for (int i = 0; i < m; ++i) {
					if (!((usem >> i) & 1)) continue;
					const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)m_h0, i), t1 = (unsigned)__builtin_amdgcn_readlane((int)m_h1, i);
					unsigned long long mask = ballot(valid && ((w0 ^ t0) | (w1 ^ t1)) == 0);
					if (!mask) continue;
					const int tb = __builtin_amdgcn_readlane(m_off, i), tlen = __builtin_amdgcn_readlane(m_len, i);
					const int ts = __builtin_amdgcn_readlane(m_ts, i);
					while (mask) {
						const int o = ob + ctz64(mask);
						mask &= mask - 1;
						const int cn = qlen - o < tlen ? qlen - o : tlen;
						consider(a32, B, cn, c0 + i, 1, o, ts, qb + o, tb, min_overlap);
					}
				}
			}
		}
	}
	// more matches, then fewer mismatches, then the earlier contig (contig.nim:32-36, :107, :239)
	if (COMBINE && G.found && (!B.b.found || G.ma > B.b.ma || (G.ma == B.b.ma && (G.mm < B.b.mm || (G.mm == B.b.mm && G.pos < B.b.pos))))) return G;
	return B.b;
}

template <class ST>
__device__ inline Best best_match_read(const ST &S, const Arena &A, const short *list, int n, int min_overlap)
{
	return best_match_all<ST, false>(S, A, ST::QSLOT, list, n, min_overlap, 0);
}

template <class ST>
__device__ inline Best best_match_combine(const ST &S, const Arena &A, int qs, const short *list, int n,
                                          int min_overlap, int max_mm)
{
	IHP_T0(A);
	const Best b = best_match_all<ST, true>(S, A, qs, list, n, min_overlap, max_mm);
	IHP_T1(A, 5);
	return b;
}

}  // namespace ihp
