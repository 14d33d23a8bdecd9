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

typedef uint32_t tally_u32u __attribute__((aligned(1)));

__device__ __forceinline__ bool lane_in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// indelope.nim:293-311: lanes = reads.  counts[0..3) = ref_support, alt_support, both_found.
// Each lane walks its read four bases per (unaligned) dword load; the byte arrays are padded so the last
// partial dword of the last read is readable.
__device__ inline void tally_reads(const uint8_t *bases, const long long *read_off, const uint8_t *mapq,
                                   long long r0, long long r1, int min_mapq, int K,
                                   unsigned long long refe, unsigned long long alte, int counts[3],
                                   int *ref_hit = nullptr, int *alt_hit = nullptr)
{
	// ref_hit / alt_hit (optional): per read, the start index of the first window that matched (-1: none), entry ri - r0
	const int lane = lane_id();
	const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1) : ~0ull;
	const int hs = 2 * (K - 1);
	int nref = 0, nalt = 0, nboth = 0;
	for (long long b = r0; b < r1; b += 64) {
		const long long ri = b + lane;
		bool rf = false, af = false;
		int rpos = -1, apos = -1;
		if (ri < r1 && !(mapq && mapq[ri] < min_mapq)) {     // :294
			const uint8_t *seq = bases + read_off[ri];
			const int n = (int)(read_off[ri + 1] - read_off[ri]);
			unsigned long long f = 0, rc = 0;
			int valid = 0;
			for (int i0 = 0; i0 < n; i0 += 4) {
				const unsigned wv = *(const tally_u32u *)(seq + i0);
				const int m = n - i0 < 4 ? n - i0 : 4;
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					if (j < m) {
						const int c = base2((uint8_t)(wv >> (8 * j)));
						if (c < 0) { valid = 0; f = rc = 0; }
						else {
							f = ((f << 2) | (unsigned long long)c) & mask;
							rc = (rc >> 2) | ((unsigned long long)(3 - c) << hs);
							if (++valid >= K) {
								const unsigned long long e = f < rc ? f : rc;
								if (e == refe && !rf) { rf = true; rpos = i0 + j - (K - 1); }   // :301-304
								if (e == alte && !af) { af = true; apos = i0 + j - (K - 1); }   // :306-309
							}
						}
					}
				}
			}
		}
		nref += popc64(ballot(rf));
		nalt += popc64(ballot(af));
		nboth += popc64(ballot(rf && af));                       // :310-311
		if (ref_hit && ri < r1) { ref_hit[ri - r0] = rpos; alt_hit[ri - r0] = apos; }
	}
	counts[0] = nref; counts[1] = nalt; counts[2] = nboth;
}

// reverse complement of a K-mer's 2-bit code
__device__ __forceinline__ unsigned long long revcomp_code(unsigned long long code, int K)
{
	unsigned long long rc = 0;
	for (int i = 0; i < K; ++i) { rc = (rc << 2) | (3ull - (code & 3ull)); code >>= 2; }
	return rc;
}

// The same tally with the reads staged through LDS: 64 reads at a time are copied with coalesced dword
// loads (they are contiguous in `bases`), then each lane walks its own read out of LDS.  A lane-per-read
// walk straight from HBM touches 64 different cache lines per load instruction.
__device__ inline void tally_reads_lds(const uint8_t *bases, const long long *read_off, const uint8_t *mapq,
                                       long long r0, long long r1, int min_mapq, int K,
                                       unsigned long long refe, unsigned long long alte, int counts[3],
                                       uint32_t *lds32, int lds_bytes, int *ref_hit = nullptr, int *alt_hit = nullptr,
                                       bool first_staged = false)
{   // first_staged: the caller has already put the first 64 reads into LDS (tally_stage_first)
	const int lane = lane_id();
	const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1) : ~0ull;
	// forward and reverse-complement codes of the two k-mers (refe/alte are the smaller of each pair)
	const unsigned long long ref_f = refe, ref_r = revcomp_code(refe, K), alt_f = alte, alt_r = revcomp_code(alte, K);
	int nref = 0, nalt = 0, nboth = 0;
	for (long long b = r0; b < r1; b += 64) {
		const long long e = b + 64 < r1 ? b + 64 : r1;
		const long long base0 = read_off[b];
		const int nbytes = (int)(read_off[e] - base0);
		if (nbytes + 8 > lds_bytes) {                            // does not fit: walk HBM directly (rare)
			int c[3];
			tally_reads(bases, read_off, mapq, b, e, min_mapq, K, refe, alte, c,
			            ref_hit ? ref_hit + (b - r0) : nullptr, alt_hit ? alt_hit + (b - r0) : nullptr);
			nref += c[0]; nalt += c[1]; nboth += c[2];
			continue;
		}
		if (!(first_staged && b == r0)) {
			WSYNC();
			for (int i = 4 * lane; i < nbytes; i += 256) lds32[i >> 2] = *(const tally_u32u *)(bases + base0 + i);
		}
		WSYNC();
		const long long ri = b + lane;
		// Branch-free walk: a window's canonical code equals the k-mer's iff its forward code equals the k-mer's
		// forward or reverse-complement code, so only the forward code is rolled and compared with four constants;
		// a base that is not upper-case ACGT zeroes the run length (:300 never sees such a window) and the K-1
		// windows it taints are masked by the run length alone.
		const bool use = ri < e && !(mapq && mapq[ri] < min_mapq);   // :294
		int off = 0, n = 0;
		if (use) { off = (int)(read_off[ri] - base0); n = (int)(read_off[ri + 1] - read_off[ri]); }
		const int nmax = wave_max_i32s(n);
		unsigned long long f = 0, rfm = 0, afm = 0;
		int run = 0, rpos = -1, apos = -1;
		for (int i0 = 0; i0 < nmax; i0 += 4) {
			const int bo = off + i0, w = bo >> 2;
			const unsigned wv = __builtin_amdgcn_alignbit(lds32[w + 1], lds32[w], (unsigned)(bo & 3) * 8u);
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const unsigned ch = (wv >> (8 * j)) & 0xffu;
				const unsigned x = (ch >> 1) & 3u, c = x ^ (x >> 1);          // A C G T -> 0 1 2 3 (and something for the rest)
				const bool acgt = __builtin_amdgcn_perm(0u, 0x54474341u, c | 0x0c0c0c00u) == ch && i0 + j < n;   // selector 0x0c: zero byte
				run = acgt ? run + 1 : 0;
				f = ((f << 2) | (unsigned long long)c) & mask;
				const unsigned long long full = ballot(run >= K);
				const unsigned long long hr = full & (ballot(f == ref_f) | ballot(f == ref_r));   // :301-309
				const unsigned long long ha = full & (ballot(f == alt_f) | ballot(f == alt_r));
				const int wpos = i0 + j - (K - 1);                            // start of this window: the first hit is kept
				rpos = lane_in_mask(hr & ~rfm) ? wpos : rpos;
				apos = lane_in_mask(ha & ~afm) ? wpos : apos;
				rfm |= hr; afm |= ha;
			}
		}
		const unsigned long long usem = ballot(use);
		rfm &= usem; afm &= usem;
		if (ref_hit && ri < e) { ref_hit[ri - r0] = use ? rpos : -1; alt_hit[ri - r0] = use ? apos : -1; }
		nref += popc64(rfm);
		nalt += popc64(afm);
		nboth += popc64(rfm & afm);                               // :310-311
	}
	counts[0] = nref; counts[1] = nalt; counts[2] = nboth;
}

// The first (usually only) group of 64 reads of a region into LDS.  Returns false when it does not fit (the tally
// then walks HBM directly).  No barrier: tally_reads_lds() waits before it reads.
__device__ __forceinline__ bool tally_stage_first(const uint8_t *bases, long long base0, long long end0, uint32_t *lds32, int lds_bytes)
{
	const int lane = lane_id();
	const int nbytes = (int)(end0 - base0);
	if (nbytes + 8 > lds_bytes) return false;
	for (int i = 4 * lane; i < nbytes; i += 256) lds32[i >> 2] = *(const tally_u32u *)(bases + base0 + i);
	return true;
}

// A k-mer code of the A C G T = 0 1 2 3 alphabet in k_prepack's A C T G = 0 1 2 3 (digit d -> d ^ (d >> 1): an involution)
__device__ __forceinline__ unsigned long long to_packed_alphabet(unsigned long long code)
{
	return code ^ ((code >> 1) & 0x5555555555555555ull);
}

// The tally on the 2-bit bases k_prepack left in HBM (read i at dword (read_off[i] >> 4) + i, from bit 0): a quarter of the
// staging of tally_reads_lds() and no per-base decoding or validation -- the caller has checked that no read of the region
// has a base that is not upper-case ACGT (k_prepack's read_bad), so every window of a read is a valid k-mer (:300).
// Lanes = reads as there; same counts, same first-hit positions.  Returns false (nothing done) when a group of 64 reads
// does not fit the LDS area.
__device__ inline bool tally_reads_packed(const uint32_t *pk, const long long *read_off, const uint8_t *mapq,
                                          long long r0, long long r1, int min_mapq, int K,
                                          unsigned long long refe, unsigned long long alte, int counts[3],
                                          uint32_t *lds32, int lds_bytes, int *ref_hit, int *alt_hit)
{
	const int lane = lane_id();
	const unsigned long long mask = K < 32 ? ((1ull << (2 * K)) - 1) : ~0ull;
	const unsigned long long ref_f = to_packed_alphabet(refe), ref_r = to_packed_alphabet(revcomp_code(refe, K));
	const unsigned long long alt_f = to_packed_alphabet(alte), alt_r = to_packed_alphabet(revcomp_code(alte, K));
	for (long long b = r0; b < r1; b += 64) {                   // fits?  (decided before anything is counted)
		const long long e = b + 64 < r1 ? b + 64 : r1;
		const long long nd = ((read_off[e] >> 4) + e) - ((read_off[b] >> 4) + b);
		if (4 * nd + 8 > lds_bytes) return false;
	}
	int nref = 0, nalt = 0, nboth = 0;
	for (long long b = r0; b < r1; b += 64) {
		const long long e = b + 64 < r1 ? b + 64 : r1;
		const long long gb = (read_off[b] >> 4) + b;
		const int nd = (int)(((read_off[e] >> 4) + e) - gb);
		WSYNC();
		for (int i = lane; i < nd; i += 64) lds32[i] = pk[gb + i];
		WSYNC();
		const long long ri = b + lane;
		const bool use = ri < e && !(mapq && mapq[ri] < min_mapq);   // :294
		int d0 = 0, n = 0;
		if (use) { const long long o = read_off[ri]; d0 = (int)((o >> 4) + ri - gb); n = (int)(read_off[ri + 1] - o); }
		const int nmax = wave_max_i32s(n);
		unsigned long long f = 0, rfm = 0, afm = 0;
		int rpos = -1, apos = -1;
		for (int j0 = 0; j0 < nmax; j0 += 16) {
			const unsigned dw = j0 < n ? lds32[d0 + (j0 >> 4)] : 0u;
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				f = ((f << 2) | (unsigned long long)((dw >> (2 * k)) & 3u)) & mask;
				const int j = j0 + k;
				if (j < K - 1) continue;                                  // (wave-uniform) no full window yet
				const unsigned long long full = ballot(j < n);
				const unsigned long long hr = full & (ballot(f == ref_f) | ballot(f == ref_r));   // :301-309
				const unsigned long long ha = full & (ballot(f == alt_f) | ballot(f == alt_r));
				const int wpos = j - (K - 1);                             // start of this window: the first hit is kept
				rpos = lane_in_mask(hr & ~rfm) ? wpos : rpos;
				apos = lane_in_mask(ha & ~afm) ? wpos : apos;
				rfm |= hr; afm |= ha;
			}
		}
		const unsigned long long usem = ballot(use);
		rfm &= usem; afm &= usem;
		if (ref_hit && ri < e) { ref_hit[ri - r0] = use ? rpos : -1; alt_hit[ri - r0] = use ? apos : -1; }
		nref += popc64(rfm);
		nalt += popc64(afm);
		nboth += popc64(rfm & afm);                               // :310-311
	}
	counts[0] = nref; counts[1] = nalt; counts[2] = nboth;
	return true;
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

constexpr int HIT_SLOTS = 4;      // tallied events per region with hit positions at a fixed place

struct TallyParams { int K, min_event_len, max_events, min_mapq_tally, fallback; };

// The CIGAR of one alignment: up to 64 words held one per lane (loaded with a single vector load, read back with
// v_readlane), longer ones from memory.  Walking a CIGAR word by word from memory is a chain of dependent loads,
// and this kernel's work items are short enough for such chains to be most of their time.
struct CigSrc {
	unsigned cw; const uint32_t *mem; bool in_lanes;
	__device__ __forceinline__ uint32_t word(int i) const { return in_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)cw, i) : mem[i]; }
};

// Ez.cigar truncation (ksw2.nim:22-33): number of events (I/D ops) among the ops the
// iterator yields; *ntrunc = number of ops yielded.
__device__ __forceinline__ int count_events(const CigSrc cigar, int n_cigar, int max_q, int *ntrunc)
{
	const uint32_t max_off = (uint32_t)max_q;
	uint32_t off = 0;
	int nev = 0, nt = 0;
	for (int i = 0; i < n_cigar; ++i) {
		if (off >= max_off) break;
		const uint32_t w = cigar.word(i), op = w & 0xf, len = w >> 4;
		if (op != 2) off += len;
		nt++;
		if (op == 1 || op == 2) nev++;
	}
	*ntrunc = nt;
	return nev;
}

// OR-reduce a 64-bit value over the wave
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v)
{
	const unsigned lo = wave_or_u32((unsigned)v), hi = wave_or_u32((unsigned)(v >> 32));
	return ((unsigned long long)hi << 32) | lo;
}

// number of distinct byte values among lanes [0, K), saturated at 3 (only "== 1" and "< 3" are ever asked)
__device__ __forceinline__ int distinct3(unsigned byte, int K)
{
	const int lane = lane_id();
	const bool in = lane < K;
	const unsigned v0 = (unsigned)bcast((int)byte, 0);
	const unsigned long long m1 = ballot(in && byte != v0);
	if (!m1) return 1;
	const unsigned v1 = (unsigned)bcast((int)byte, ctz64(m1));
	return ballot(in && byte != v0 && byte != v1) ? 3 : 2;
}

// canonical 2-bit code of the k-mer held one byte per lane; false if a byte is not upper-case ACGT
__device__ __forceinline__ bool mincode_lanes(unsigned byte, int K, unsigned long long &code)
{
	const int lane = lane_id();
	const int b = lane < K ? base2((uint8_t)byte) : 0;
	if (ballot(lane < K && b < 0)) return false;
	unsigned long long f = 0, rc = 0;
	if (lane < K) { f = (unsigned long long)b << (2 * (K - 1 - lane)); rc = (unsigned long long)(3 - b) << (2 * lane); }
	f = wave_or_u64(f); rc = wave_or_u64(rc);
	code = f < rc ? f : rc;
	return true;
}

// Events of one alignment (the caller has checked 0 < nev <= max_events, indelope.nim:229).
// ctg_rel = ctg.start - region origin; reference points at the window the contig was
// aligned to (length reflen).  The ref/alt k-mers are held one byte per lane (K <= 31).
__device__ inline void fill_events(const CigSrc cigar, int ntrunc,
                                   const uint8_t *ctg, int ctg_len, int ctg_rel,
                                   const uint8_t *reference, int reflen,
                                   const uint8_t *bases, const long long *read_off, const uint8_t *mapq,
                                   long long r0, long long r1, const TallyParams P, DevEvent *ev,
                                   uint32_t *lds32, int lds_bytes,
                                   int job, int ev_index0, FbItem *fb_items, int *fb_count,
                                   int *hit_pool, unsigned long long *hit_cursor, long long hit_cap, int *hit_overflow,
                                   int *region_cnt, long long region_base, long long bump0,
                                   long long base0, long long end0, const uint32_t *pk = nullptr)
{   // pk: the region's reads as 2-bit bases (all of them upper-case ACGT), or null.  base0 / end0: read_off of the region's first read and of the end of its first group of 64 (loaded by the caller)
	const int lane = lane_id();
	const int K = P.K;
	const int width = (int)((double)(K + 1) / 2.0 - 1.0);                // :218
	int toff = ctg_rel, qoff = 0, ii = -1;
	bool staged = false;
	for (int i = 0; i < ntrunc; ++i) {
		const uint32_t cwi = cigar.word(i), op = cwi & 0xf, len = cwi >> 4;
		if (op == 0) { toff += (int)len; qoff += (int)len; continue; }
		++ii;
		int e_type, e_ts, e_te, e_qs, e_qe;
		if (op == 1) {                                                   // ksw2.nim:75-76, :88-89
			e_type = 0; e_ts = toff; e_te = toff + 1; e_qs = qoff; e_qe = qoff + (int)len;
			qoff += (int)len;
		} else {                                                         // ksw2.nim:77-78, :86-87
			e_type = 1; e_ts = toff; e_te = toff + (int)len; e_qs = qoff; e_qe = qoff + 1;
			toff += (int)len;
		}
		int status = -1, cf = 0;
		unsigned rk = 0, ak = 0;                                         // this lane's byte of ref_kmer / alt_kmer
		unsigned long long refe = 0, alte = 0;
		int counts[3] = {0, 0, 0};
		if ((int)len < P.min_event_len) status = IHP_EV_SHORT;           // :234
		else if (reflen < K || ctg_len < K) status = IHP_EV_OOB;
		else {
			int tstart = e_ts - ctg_rel - width;                         // :236-238
			if (tstart < 0) tstart = 0;
			if (tstart + K > reflen) tstart = reflen - K;
			if (lane < K) rk = reference[tstart + lane];                 // :240
			const int o1 = e_qs, o2 = ctg_len - e_qe - 1;
			cf = o1 < o2 ? o1 : o2;                                      // :243
			int qstart = e_qs - width;                                   // :244-246
			if (qstart < 0) qstart = 0;
			if (qstart + K > ctg_len) qstart = ctg_len - K;
			if (lane < K) ak = ctg[qstart + lane];                       // :248
			// the reads travel to LDS while the k-mer bytes are still on their way
			if (!pk && !staged && r1 > r0) staged = tally_stage_first(bases, base0, end0, lds32, lds_bytes);
			if (!ballot(lane < K && rk != ak)) {                         // :255-262
				qstart = e_qs - 3;
				if (qstart < 0) qstart = 0;
				if (qstart + K > ctg_len) {
					const int qend = e_qe + 4 < ctg_len ? e_qe + 4 : ctg_len;
					if (qend - K < 0) status = IHP_EV_OOB;
					else if (lane < K) ak = ctg[qend - K + lane];
				} else if (lane < K) ak = ctg[qstart + lane];
			}
			if (status < 0) {
				const bool same = !ballot(lane < K && rk != ak);
				if (same && (e_qs == 0 || distinct3(ak, K) == 1)) status = IHP_EV_SAME_KMER;   // :264
				else if (distinct3(rk, K) < 3) status = IHP_EV_LOW_CPLX;                       // :266
				else if (same) status = IHP_EV_BUG_SAME;                                       // :268-275
				else if (!mincode_lanes(rk, K, refe) || !mincode_lanes(ak, K, alte)) status = IHP_EV_NON_ACGT;
			}
		}
		long long hoff = -1;
		if (status < 0) {
			// first-hit positions of every read (the side data of :302-309): 2 x nreads ints from a bump pool
			const long long nr = r1 - r0;
			if (hit_pool && nr > 0) {
				// the region's own slots first (HIT_SLOTS events of 2 x nr ints at 8 x its first read index: one atomic on
				// a per-region word); further events take space from the shared bump region behind them.  A single bump
				// cursor for everything serialises ten thousand waves on one address.
				int k = HIT_SLOTS;
				if (region_cnt) { if (lane == 0) k = atomicAdd(region_cnt, 1); k = uni(k); }
				if (k < HIT_SLOTS) hoff = region_base + 2 * nr * k;
				else {
					long long o = 0;
					if (lane == 0) o = (long long)atomicAdd(hit_cursor, (unsigned long long)(2 * nr));
					hoff = bump0 + uni(o);
					if (hoff + 2 * nr > hit_cap) { hoff = -1; if (lane == 0) atomicExch(hit_overflow, 1); }
				}
			}
			if (!(pk && tally_reads_packed(pk, read_off, mapq, r0, r1, P.min_mapq_tally, K, refe, alte, counts, lds32, lds_bytes,
			                               hoff >= 0 ? hit_pool + hoff : nullptr, hoff >= 0 ? hit_pool + hoff + nr : nullptr))) {
				tally_reads_lds(bases, read_off, mapq, r0, r1, P.min_mapq_tally, K, refe, alte, counts, lds32, lds_bytes,
				                hoff >= 0 ? hit_pool + hoff : nullptr, hoff >= 0 ? hit_pool + hoff + nr : nullptr, staged);
				if (r1 - r0 > 64) staged = false;                        // later groups have overwritten the first one
			} else staged = false;
			status = IHP_EV_TALLIED;
		}
		DevEvent *o = ev + ii;
		if (lane == 0) {
			o->tstart_rel = e_ts; o->tstop_rel = e_te; o->qstart = e_qs; o->qstop = e_qe;
			o->len = len; o->type = (unsigned char)e_type; o->status = (unsigned char)status;
			const bool fb = status == IHP_EV_TALLIED && counts[2] > 0;   // :313
			const bool run_fb = fb && P.fallback && fb_items;
			o->fallback = fb;
			o->aligned = run_fb; o->cf_offset = cf;                      // :372
			o->kmer_ref = counts[0]; o->kmer_alt = counts[1]; o->kmer_both = counts[2];
			o->hit_off = hoff;
			// the fallback kernel counts its votes into the zeroed fields (:316, :320-321)
			o->ref_support = run_fb ? 0 : counts[0]; o->alt_support = run_fb ? 0 : counts[1];
			o->both_found = run_fb ? 0 : counts[2];
			if (run_fb) {
				const int k = atomicAdd(fb_count, 1);
				fb_items[k].job = job; fb_items[k].ev = ev_index0 + ii;
			}
		}
		if (lane < 32) { o->ref_kmer[lane] = (char)(lane < K ? rk : 0); o->alt_kmer[lane] = (char)(lane < K ? ak : 0); }
	}
}

}  // namespace ihp
