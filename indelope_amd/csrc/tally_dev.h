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
// Round 6: no rolling code.  The packed read IS the sequence of its windows -- the window that starts at base p is 2 K bits of the
// stream from bit 2 p, first base lowest -- so the four k-mer codes are turned into that order once, a window's first sixteen bases
// are one v_alignbit of two neighbouring dwords, and a position costs that and four 32-bit compares against scalars; only a
// position at which some read's first sixteen bases match goes on to the rest of the window, the end of the read and the
// first-hit bookkeeping (one position in six on C2).  It was a 64-bit shift, mask and four 64-bit compares per base with all
// of the bookkeeping every time: 12 VALU + 11 SALU a base, now 5 + 4.
// One group of up to 64 packed reads -> LDS: dwords [gb, gb + nd) of pk by the loads that write LDS themselves (global_load_lds:
// a wave's 64 dwords land at consecutive addresses from a wave-uniform base, no register in between), all of them in flight at
// once; whoever reads the area waits with WSYNC.  (A dword a lane through a register and a wait per round was eleven dependent
// round trips to HBM for 64 reads of 150 bases -- most of a job's time; sixteen bytes a lane through registers cost 35 spilled
// registers in a kernel that keeps eight waves per SIMD.)
__device__ __forceinline__ void tally_stage_packed(const uint32_t *src, int nd, uint32_t *lds32)
{
	typedef __attribute__((address_space(3))) uint32_t *lds_p;
	typedef const __attribute__((address_space(1))) uint32_t *glb_p;
	const int lane = lane_id();
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (what was read from the area before has been read)
	for (int i0 = 0; i0 < nd; i0 += 64)
		if (i0 + lane < nd) __builtin_amdgcn_global_load_lds((glb_p)(src + i0 + lane), (lds_p)(lds32 + i0), 4, 0, 0);
}
// A lane's read of the group [b, e): first dword relative to the group's and length; n = 0: not used (:294) or no read
__device__ __forceinline__ void tally_lane_read(const long long *read_off, const uint8_t *mapq, long long b, long long e, long long gb, int min_mapq, int &d0, int &n)
{
	const long long ri = b + lane_id();
	const long long rc = ri < e ? ri : e - 1;                       // (every lane loads: nothing waits on a branch)
	const long long o = read_off[rc], o1 = read_off[rc + 1];
	const bool use = ri < e && !(mapq && mapq[rc] < min_mapq);     // :294
	d0 = use ? (int)((o >> 4) + ri - gb) : 0; n = use ? (int)(o1 - o) : 0;
}
// The first group of a region, ahead of the tally (fill_events: while the k-mer bytes are on their way).  base0 / end0: read_off of
// the region's first read and of the end of its first group.  False: it does not fit the LDS area.
__device__ __forceinline__ bool tally_stage_packed_first(const uint32_t *pk, const long long *read_off, const uint8_t *mapq, long long r0, long long r1,
                                                         long long base0, long long end0, int min_mapq, uint32_t *lds32, int lds_bytes, int &d0, int &n)
{
	const long long e = r0 + 64 < r1 ? r0 + 64 : r1;
	const long long gb = (base0 >> 4) + r0;
	const int nd = (int)(((end0 >> 4) + e) - gb);
	if (4 * nd + 8 > lds_bytes) return false;
	tally_stage_packed(pk + gb, nd, lds32);
	tally_lane_read(read_off, mapq, r0, e, gb, min_mapq, d0, n);   // (behind the staging loads: its answers are looked at right away)
	return true;
}
__device__ __forceinline__ unsigned long long kmer_stream_order(unsigned long long code, int K)
{   // code: first base in the highest of its 2 K bits (the rolling order) -> first base in the lowest
	unsigned long long x = __builtin_bitreverse64(code);            // bases in reverse order, and the two bits of every base swapped
	x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
	return x >> (64 - 2 * K);
}
template <bool KGE16>                                           // K >= 16 (the usual 27): a window's first sixteen bases need no mask
__device__ inline bool tally_reads_packed_k(const uint32_t *pk, const long long *read_off, const uint8_t *mapq,
                                            long long r0, long long r1, int min_mapq, int K,
                                            unsigned long long refe, unsigned long long alte, int counts[3],
                                            uint32_t *lds32, int lds_bytes, int *ref_hit, int *alt_hit, bool first_staged, int fd0, int fn)
{   // first_staged: the caller has put the first group into LDS (tally_stage_packed_first; fd0 / fn: what it returned)
	const int lane = lane_id();
	const unsigned long long t0 = kmer_stream_order(to_packed_alphabet(refe), K), t1 = kmer_stream_order(to_packed_alphabet(revcomp_code(refe, K)), K);
	const unsigned long long t2 = kmer_stream_order(to_packed_alphabet(alte), K), t3 = kmer_stream_order(to_packed_alphabet(revcomp_code(alte, K)), K);
	const unsigned mlo = K >= 16 ? 0xffffffffu : (1u << (2 * K)) - 1u;                       // a window's first sixteen bases ...
	const unsigned mhi = K <= 16 ? 0u : K >= 32 ? 0xffffffffu : (1u << (2 * K - 32)) - 1u;   // ... and the rest of it
	const unsigned l0 = (unsigned)t0, l1 = (unsigned)t1, l2 = (unsigned)t2, l3 = (unsigned)t3;
	const unsigned h0 = (unsigned)(t0 >> 32), h1 = (unsigned)(t1 >> 32), h2 = (unsigned)(t2 >> 32), h3 = (unsigned)(t3 >> 32);
	for (long long b = first_staged ? r0 + 64 : r0; b < r1; b += 64) {   // fits?  (decided before anything is counted)
		const long long e = b + 64 < r1 ? b + 64 : r1;
		const long long nd = ((read_off[e] >> 4) + e) - ((read_off[b] >> 4) + b);
		if (4 * nd + 8 > lds_bytes) return false;
	}
	int nref = 0, nalt = 0, nboth = 0;
	for (long long b = r0; b < r1; b += 64) {
		const long long e = b + 64 < r1 ? b + 64 : r1;
		const long long ri = b + lane;
		int d0 = fd0, n = fn;
		if (!(first_staged && b == r0)) {
			const long long gb = (read_off[b] >> 4) + b;
			const int nd = (int)(((read_off[e] >> 4) + e) - gb);
			tally_stage_packed(pk + gb, nd, lds32);
			tally_lane_read(read_off, mapq, b, e, gb, min_mapq, d0, n);
		}
		WSYNC();
		const int nmax = wave_max_i32s(n);
		const int nq = ((n + 15) >> 4) + 1;                           // dwords 0 .. nq from the read's first are the read's own or the two behind them (staged or slack)
		unsigned long long rfm = 0, afm = 0;
		int rpos = -1, apos = -1;
		unsigned w0 = lds32[d0], w1 = lds32[d0 + 1];
		for (int p0 = 0; p0 + K <= nmax; p0 += 16) {                  // window starts p0 .. p0 + 15
			const int q = (p0 >> 4) + 2;
			const unsigned w2 = q <= nq ? lds32[d0 + q] : 0u;
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				unsigned lo = k ? __builtin_amdgcn_alignbit(w1, w0, 2u * (unsigned)k) : w0;
				if (!KGE16) lo &= mlo;
				const unsigned long long c0 = ballot(lo == l0), c1 = ballot(lo == l1), c2 = ballot(lo == l2), c3 = ballot(lo == l3);
				if (!((c0 | c1) | (c2 | c3))) continue;                  // the usual position: nobody's window starts like a k-mer
				const int p = p0 + k;
				const unsigned hi = (k ? __builtin_amdgcn_alignbit(w2, w1, 2u * (unsigned)k) : w1) & mhi;
				const unsigned long long full = ballot(p + K <= n);
				const unsigned long long hr = full & ((c0 & ballot(hi == h0)) | (c1 & ballot(hi == h1)));   // :301-309
				const unsigned long long ha = full & ((c2 & ballot(hi == h2)) | (c3 & ballot(hi == h3)));
				rpos = lane_in_mask(hr & ~rfm) ? p : rpos;                // the first hit is kept
				apos = lane_in_mask(ha & ~afm) ? p : apos;
				rfm |= hr; afm |= ha;
			}
			w0 = w1; w1 = w2;
		}
		// (a read that is not used, :294, has n = 0 here: no window of it is full, its bits stay clear and its positions -1)
		if (ref_hit && ri < e) { ref_hit[ri - r0] = rpos; alt_hit[ri - r0] = apos; }
		nref += popc64(rfm);
		nalt += popc64(afm);
		nboth += popc64(rfm & afm);                               // :310-311
	}
	counts[0] = nref; counts[1] = nalt; counts[2] = nboth;
	return true;
}
__device__ inline bool tally_reads_packed(const uint32_t *pk, const long long *read_off, const uint8_t *mapq,
                                          long long r0, long long r1, int min_mapq, int K,
                                          unsigned long long refe, unsigned long long alte, int counts[3],
                                          uint32_t *lds32, int lds_bytes, int *ref_hit, int *alt_hit, bool first_staged = false, int fd0 = 0, int fn = 0)
{
	if (K >= 16) return tally_reads_packed_k<true>(pk, read_off, mapq, r0, r1, min_mapq, K, refe, alte, counts, lds32, lds_bytes, ref_hit, alt_hit, first_staged, fd0, fn);
	return tally_reads_packed_k<false>(pk, read_off, mapq, r0, r1, min_mapq, K, refe, alte, counts, lds32, lds_bytes, ref_hit, alt_hit, first_staged, fd0, fn);
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
	bool pstaged = false; int pd0 = 0, pn = 0;                          // the packed reads' first group is in LDS (lane: first dword, length)
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
			if (pk && !pstaged && r1 > r0) pstaged = tally_stage_packed_first(pk, read_off, mapq, r0, r1, base0, end0, P.min_mapq_tally, lds32, lds_bytes, pd0, pn);
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
			// (a first group that did not fit has left nothing staged: both kinds of staging are decided before they write)
			if (!(pk && (pstaged || r1 <= r0) && tally_reads_packed(pk, read_off, mapq, r0, r1, P.min_mapq_tally, K, refe, alte, counts, lds32, lds_bytes,
			                               hoff >= 0 ? hit_pool + hoff : nullptr, hoff >= 0 ? hit_pool + hoff + nr : nullptr, pstaged, pd0, pn))) {
				tally_reads_lds(bases, read_off, mapq, r0, r1, P.min_mapq_tally, K, refe, alte, counts, lds32, lds_bytes,
				                hoff >= 0 ? hit_pool + hoff : nullptr, hoff >= 0 ? hit_pool + hoff + nr : nullptr, staged && !pstaged);
				if (r1 - r0 > 64) staged = false;                        // later groups have overwritten the first one
				pstaged = false;
			} else { staged = false; if (r1 - r0 > 64) pstaged = false; }
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
