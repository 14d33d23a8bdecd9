// asm2_dev.h -- read phase of assemble (indelope.nim:163-169 over contig.nim:70-141, :156-252) on 2-bit packed bases.
//
// k_assemble's first pass spent 10 us per read in ONE dependent chain (22 000 cycles for a wave alone on its CU:
// tools/r2_probe.sh) -- contig metadata gathered from LDS for every read, a scalar walk to map lanes to contigs, LDS
// crossbar shuffles, HBM round trips for the support difference arrays.  This version keeps the whole contig
// directory in registers (lane c = contig c), the bases 2 bits each in LDS (16 per dword), the lane -> (contig, dword)
// work list in a register that only changes when a contig grows, and records per read where it went instead of
// touching per-base supports at all (they are counted once at the end).  A fresh read (support 1, nreads 1) can only
// match exactly (contig.nim:44-47 cannot fire: every contig base has support >= 1), so slide_align is "longest exactly
// matching overlap, first in scan order" and a 16-base window (one dword) is an exact prefilter:
//   target offsets (contig.nim:81-111): a lane takes one dword of one contig and tests its 16 windows against the read's
//     first 16 bases (16 x v_alignbit + v_xor, folded with v_min3);
//   query offsets (:114-135): lane o holds the read's window at offset o, the contigs' first dwords come from the
//     directory one v_readlane each.
// Survivors are verified on the whole overlap (16 bases per lane) and ranked under the reference's total order
// (matches desc, contig index asc, target phase before query phase, offset asc).
// Preconditions (anything else is forwarded to the byte-based passes, whose results are identical): every eligible read
// is upper-case ACGT with 20 <= trimmed length <= 1008, max_mismatch == 0, 0 < min_overlap_pct <= 1, at most 64 contigs.
#pragma once
#include "contig_dev.h"

namespace ihp {

typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));

// ------------------------------------------------------------------------------------------------ packing
// code = (c >> 1) & 3: A 0, C 1, T 2, G 3; "ACTG"[code] gives the base back
constexpr unsigned PK_LUT = 0x47544341u;
__device__ __forceinline__ unsigned pack4(unsigned w, unsigned &diff)
{   // four ASCII bases -> 8 bits (base k at bits 2k); diff |= bytes that are not one of ACGT
	const unsigned sel = (w >> 1) & 0x03030303u;
	diff |= __builtin_amdgcn_perm(0u, PK_LUT, sel) ^ w;
	const unsigned x = sel | (sel >> 6);
	return (x | (x >> 12)) & 0xffu;
}

struct PrepackArgs {
	long long n_reads;
	const long long *read_off;
	const uint8_t *bases, *quals;             // quals may be null
	const uint8_t *bases4;                    // or null: the bases 4 bits each as BAM stores them (read i from byte (read_off[i] >> 1) + i, first base in
	uint8_t *bases_w;                         //   the high nibble); their ASCII form ("=ACMGRSVTWYHKDBN") is then written to bases_w (= bases)
	const int *trim_lo_in, *trim_hi_in;       // the stager's trim bounds, or null
	int trim_min_qual;
	uint32_t *pk;                             // read i at dword (read_off[i] >> 4) + i, ceil(len / 16) dwords, tail bits zero
	int *trim_lo, *trim_hi;                   // [n_reads] out: kept range [lo, hi) of every read (indelope.nim:23-38)
	uint8_t *read_bad;                        // [n_reads] out: 1 = a base that is not upper-case ACGT
	unsigned long long *t_start;              // optional: device wall clock when the launch starts to execute (see mark_start())
};

// Sixteen bases, 4 bits each as BAM stores them (n0 = the first eight, first base in the high nibble of the first byte), as ASCII
// ("=ACMGRSVTWYHKDBN") into w and -- the `rem` of them that belong to the read -- to o.
__device__ __forceinline__ void b4_to_ascii(unsigned n0, unsigned n1, int rem, uint8_t *o, unsigned (&w)[4])
{
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const unsigned x = ((k < 2 ? n0 : n1) >> (16 * (k & 1))) & 0xffffu;          // two bytes = four bases
		const unsigned sel = ((x >> 4) & 0xfu) | ((x & 0xfu) << 8) | (((x >> 12) & 0xfu) << 16) | (((x >> 8) & 0xfu) << 24);
		const unsigned lo8 = __builtin_amdgcn_perm(0x56535247u, 0x4d43413du, sel & 0x07070707u);   // "=ACM" "GRSV"
		const unsigned hi8 = __builtin_amdgcn_perm(0x4e42444bu, 0x48595754u, sel & 0x07070707u);   // "TWYH" "KDBN"
		const unsigned m = ((sel >> 3) & 0x01010101u) * 0xffu;
		w[k] = (lo8 & ~m) | (hi8 & m);
		const int v = rem - 4 * k;
		if (v >= 4) *(u32_unaligned_t *)(o + 4 * k) = w[k];
		else for (int j = 0; j < v; ++j) o[4 * k + j] = (uint8_t)(w[k] >> (8 * j));
	}
}

// One 16-lane group per read (four reads per wave at a time), grid-stride over all reads of the batch.
__global__ __launch_bounds__(64) void k_prepack(const PrepackArgs a)
{
	const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
	if (a.t_start && blockIdx.x == 0 && threadIdx.x == 0) *a.t_start = (unsigned long long)wall_clock64();
	const long long stride = (long long)gridDim.x * 4;
	for (long long i0 = (long long)blockIdx.x * 4; i0 < a.n_reads; i0 += stride) {
		const long long ri = i0 + grp;
		const bool live = ri < a.n_reads;
		long long off = 0; int len = 0;
		if (live) { off = a.read_off[ri]; len = (int)(a.read_off[ri + 1] - off); }
		// ---- trim bounds
		int lo = 0, hi = len;
		if (a.trim_lo_in) {
			if (live) {
				lo = a.trim_lo_in[ri]; hi = a.trim_hi_in[ri];
				lo = lo < 0 ? 0 : lo > len ? len : lo;
				hi = hi > len ? len : hi; hi = hi < lo ? lo : hi;
			}
		} else if (a.quals) {
			// indelope.nim:23-38 inside a 16-lane group: a = first i < high with q >= min, else high; b = last i > a with q >= min
			const int high = len - 1;
			const unsigned mq = (unsigned)a.trim_min_qual & 0xff;
			int aa = high > 0 ? high : 0;
			bool found = !(live && high > 0);
			for (int b0 = 0; ; b0 += 16) {
				const int i = b0 + sub;
				const bool g = !found && i < high && a.quals[off + i] >= mq;
				const unsigned long long m = ballot(g);
				const unsigned gm = (unsigned)(m >> (16 * grp)) & 0xffffu;
				if (!found && gm) { aa = b0 + __builtin_ctz(gm); found = true; }
				if (!found && b0 + 16 >= high) found = true;
				if (!ballot(!found)) break;
			}
			if (aa == high || len <= 0) { lo = 0; hi = 0; }
			else {
				int bb = aa;
				bool done = !live;
				for (int top = high; ; top -= 16) {
					const int i = top - sub;
					const bool g = !done && i > aa && a.quals[off + i] >= mq;
					const unsigned long long m = ballot(g);
					const unsigned gm = (unsigned)(m >> (16 * grp)) & 0xffffu;
					if (!done && gm) { bb = top - __builtin_ctz(gm); done = true; }
					if (!done && top - 16 <= aa) done = true;
					if (!ballot(!done)) break;
				}
				lo = aa; hi = bb + 1;
			}
			// groups leave the loops together (wave-uniform exits); a dead or short group just idles
		} else if (len == 1) hi = 0;                         // no qualities = all 255: trim() still empties a 1-base read (:28-30)
		if (live && sub == 0) { a.trim_lo[ri] = lo; a.trim_hi[ri] = hi; }
		// ---- pack
		unsigned diff = 0;
		const long long pkb = (off >> 4) + ri;
		const int nd = (len + 15) >> 4;
		for (int d = sub; d < nd; d += 16) {
			unsigned w[4];
			const int rem = len - 16 * d;                        // valid bases in this dword (>= 1)
			if (a.bases4) {
				const uint8_t *q = a.bases4 + (off >> 1) + ri + 8 * d;
				const unsigned n0 = *(const u32_unaligned_t *)q, n1 = *(const u32_unaligned_t *)(q + 4);
				b4_to_ascii(n0, n1, rem, a.bases_w + off + 16 * d, w);
			} else {
				const uint8_t *p = a.bases + off + 16 * d;
#pragma unroll
				for (int k = 0; k < 4; ++k) w[k] = *(const u32_unaligned_t *)(p + 4 * k);
			}
			unsigned out = 0, df = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				unsigned dk = 0;
				const unsigned y = pack4(w[k], dk);
				const int v = rem - 4 * k;                       // valid bases of this group of four
				if (v < 4) dk &= v <= 0 ? 0u : (1u << (8 * v)) - 1u;
				df |= dk;
				out |= y << (8 * k);
			}
			if (rem < 16) out &= (1u << (2 * rem)) - 1u;
			a.pk[pkb + d] = out;
			diff |= df;
		}
		const unsigned long long bm = ballot(diff != 0);
		if (live && sub == 0) a.read_bad[ri] = (uint8_t)(((bm >> (16 * grp)) & 0xffffull) != 0);
	}
}

// The same for the usual batch -- ASCII bases, trim bounds from the stager -- as a software pipeline.  k_prepack's iteration is
// two dependent round trips to HBM (a read's offsets, then its bases) with four reads of 150 bytes in flight per wave: at
// 32 waves per CU and ~3 us per iteration that is the 1.5 TB/s it ran at, whatever the grid.  Here a wave has the bases of
// iteration i+1 and the offsets and bounds of iteration i+2 in flight while it packs iteration i.
// Every load of the loop is unconditional (indices clamped into the arrays) and nothing is computed from a loaded value before
// the top of the next iteration: the loop has ONE wait, and what it waits for has had a whole iteration to arrive.
struct PrepackRaw { long long off0, off1; };
__device__ __forceinline__ PrepackRaw prepack_raw(const PrepackArgs &a, long long ri)
{
	const long long rc = ri < a.n_reads ? ri : a.n_reads - 1;
	PrepackRaw m;
	m.off0 = a.read_off[rc]; m.off1 = a.read_off[rc + 1];
	return m;
}
struct PrepackRead { long long off; int len; };
__device__ __forceinline__ PrepackRead prepack_read(const PrepackRaw &m) { PrepackRead r; r.off = m.off0; r.len = (int)(m.off1 - m.off0); return r; }
struct PrepackData { unsigned w[4]; };
template <bool B4>
__device__ __forceinline__ PrepackData prepack_data(const PrepackArgs &a, const PrepackRead &m, long long ri, int sub)
{   // dwords [4 sub, 4 sub + 4) of the read's bytes (lanes past the read load its first ones; like k_prepack, the last group may
	// reach up to 15 bytes past the read: masked when it is packed); B4: the eight bytes that hold those sixteen bases in w[0..1]
	PrepackData d;
	if (B4) {
		const long long rc = ri < a.n_reads ? ri : a.n_reads - 1;
		const uint8_t *q = a.bases4 + (m.off >> 1) + rc + (16 * sub < m.len ? 8 * sub : 0);
		d.w[0] = *(const u32_unaligned_t *)q; d.w[1] = *(const u32_unaligned_t *)(q + 4); d.w[2] = d.w[3] = 0;
		return d;
	}
	const uint8_t *p = a.bases + m.off + (16 * sub < m.len ? 16 * sub : 0);
#pragma unroll
	for (int k = 0; k < 4; ++k) d.w[k] = *(const u32_unaligned_t *)(p + 4 * k);
	return d;
}
__device__ __forceinline__ unsigned prepack_words(const unsigned (&w)[4], int rem, unsigned &diff)
{   // sixteen ASCII bases -> one dword; rem: how many of them belong to the read (>= 1)
	unsigned out = 0, df = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		unsigned dk = 0;
		const unsigned y = pack4(w[k], dk);
		const int v = rem - 4 * k;
		if (v < 4) dk &= v <= 0 ? 0u : (1u << (8 * v)) - 1u;
		df |= dk;
		out |= y << (8 * k);
	}
	if (rem < 16) out &= (1u << (2 * rem)) - 1u;
	diff |= df;
	return out;
}

// U reads per 16-lane group and iteration (4 U per wave).  The kept ranges are a map over the reads of their own (lane <-> read,
// coalesced) in front of the pipeline, which then carries a read's offset and length and nothing else.
// B4: the bases come 4 bits each (a batch uploaded as a slab) and their ASCII form is written on the way, as k_prepack does.
template <int U, bool B4 = false>
__global__ __launch_bounds__(64) void k_prepack_fast(const PrepackArgs a)
{
	const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
	if (a.t_start && blockIdx.x == 0 && threadIdx.x == 0) *a.t_start = (unsigned long long)wall_clock64();
	for (long long ri = (long long)blockIdx.x * 64 + lane; ri < a.n_reads; ri += (long long)gridDim.x * 64) {
		const int len = (int)(a.read_off[ri + 1] - a.read_off[ri]);
		int lo = a.trim_lo_in[ri], hi = a.trim_hi_in[ri];
		lo = lo < 0 ? 0 : lo > len ? len : lo;
		hi = hi > len ? len : hi; hi = hi < lo ? lo : hi;
		a.trim_lo[ri] = lo; a.trim_hi[ri] = hi;
	}
	const long long stride = (long long)gridDim.x * 4 * U;
	long long i0 = (long long)blockIdx.x * 4 * U;
	if (i0 >= a.n_reads) return;
	PrepackRead cur[U], nxt[U];
	PrepackRaw nn[U];
	PrepackData dcur[U];
#pragma unroll
	for (int u = 0; u < U; ++u) cur[u] = prepack_read(prepack_raw(a, i0 + 4 * u + grp));
#pragma unroll
	for (int u = 0; u < U; ++u) { nn[u] = prepack_raw(a, i0 + stride + 4 * u + grp); dcur[u] = prepack_data<B4>(a, cur[u], i0 + 4 * u + grp, sub); }
	for (; i0 < a.n_reads; i0 += stride) {
		PrepackData dnxt[U];
#pragma unroll
		for (int u = 0; u < U; ++u) { nxt[u] = prepack_read(nn[u]); dnxt[u] = prepack_data<B4>(a, nxt[u], i0 + stride + 4 * u + grp, sub); }
#pragma unroll
		for (int u = 0; u < U; ++u) nn[u] = prepack_raw(a, i0 + 2 * stride + 4 * u + grp);
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const long long ri = i0 + 4 * u + grp;
			const bool live = ri < a.n_reads;
			const int len = live ? cur[u].len : 0;
			unsigned diff = 0;
			const long long pkb = (cur[u].off >> 4) + ri;
			const int nd = (len + 15) >> 4;
			if (sub < nd) {
				if (B4) {
					unsigned w[4];
					b4_to_ascii(dcur[u].w[0], dcur[u].w[1], len - 16 * sub, a.bases_w + cur[u].off + 16 * sub, w);
					a.pk[pkb + sub] = prepack_words(w, len - 16 * sub, diff);
				} else a.pk[pkb + sub] = prepack_words(dcur[u].w, len - 16 * sub, diff);
			}
			if (nd > 16) {                                       // reads longer than 256 bases: the rest as it comes
				for (int d = sub + 16; d < nd; d += 16) {
					unsigned w[4];
					if (B4) {
						const uint8_t *q = a.bases4 + (cur[u].off >> 1) + ri + 8 * d;
						b4_to_ascii(*(const u32_unaligned_t *)q, *(const u32_unaligned_t *)(q + 4), len - 16 * d, a.bases_w + cur[u].off + 16 * d, w);
					} else {
						const uint8_t *p = a.bases + cur[u].off + 16 * d;
#pragma unroll
						for (int k = 0; k < 4; ++k) w[k] = *(const u32_unaligned_t *)(p + 4 * k);
					}
					a.pk[pkb + d] = prepack_words(w, len - 16 * d, diff);
				}
			}
			const unsigned long long bm = ballot(diff != 0);
			if (live && sub == 0) a.read_bad[ri] = (uint8_t)(((bm >> (16 * grp)) & 0xffffull) != 0);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) { cur[u] = nxt[u]; dcur[u] = dnxt[u]; }
	}
}

// A batch whose compact slab brought the reads 2 bits each in this very form (IHP_SLAB2_BASES_2BIT): nothing to pack -- the kept
// ranges, the (all clear) "not ACGT" flags and the ASCII copy of the bases that the byte-based kernels, the fallback and the
// plain tally read.  One 16-lane group per read, a lane per packed word.
__global__ __launch_bounds__(64) void k_unpack_pk(const PrepackArgs a)
{
	const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
	if (a.t_start && blockIdx.x == 0 && threadIdx.x == 0) *a.t_start = (unsigned long long)wall_clock64();
	for (long long ri = (long long)blockIdx.x * 64 + lane; ri < a.n_reads; ri += (long long)gridDim.x * 64) {
		const int len = (int)(a.read_off[ri + 1] - a.read_off[ri]);
		int lo = a.trim_lo_in ? a.trim_lo_in[ri] : 0, hi = a.trim_hi_in ? a.trim_hi_in[ri] : (len == 1 ? 0 : len);
		lo = lo < 0 ? 0 : lo > len ? len : lo;
		hi = hi > len ? len : hi; hi = hi < lo ? lo : hi;
		a.trim_lo[ri] = lo; a.trim_hi[ri] = hi; a.read_bad[ri] = 0;
	}
	const long long stride = (long long)gridDim.x * 4;
	for (long long i0 = (long long)blockIdx.x * 4; i0 < a.n_reads; i0 += stride) {
		const long long ri = i0 + grp;
		if (ri >= a.n_reads) continue;
		const long long off = a.read_off[ri];
		const int len = (int)(a.read_off[ri + 1] - off);
		const uint32_t *src = a.pk + (off >> 4) + ri;
		uint8_t *dst = a.bases_w + off;
		for (int d = sub; 16 * d < len; d += 16) {
			const unsigned w = src[d];
#pragma unroll
			for (int q = 0; q < 4; ++q) {                            // four bases: 2-bit codes -> "ACTG" bytes
				const unsigned c8 = (w >> (8 * q)) & 0xffu, t = c8 | (c8 << 6), u = t | (t << 12);
				const unsigned x = __builtin_amdgcn_perm(0u, PK_LUT, u & 0x03030303u);
				const int at = 16 * d + 4 * q, v = len - at;
				if (v >= 4) *(u32_unaligned_t *)(dst + at) = x;
				else for (int j = 0; j < v; ++j) dst[at + j] = (uint8_t)(x >> (8 * j));
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------ bit helpers
// bits [sh, sh + 32) of the 64-bit value hi:lo (sh in 0..31)
__device__ __forceinline__ unsigned fsh(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// Does any of the 16 windows [j, j + 16), j = 0..15, of the 32 bases w1:w0 equal qh?  (16 x v_alignbit + v_xor folded with
// v_min3; which ones is worked out for the rare hit lanes only, by window_bits.)
// Round 6: as four QUAD-BYTE SADs.  A 16-base window is four bytes; window j = 4 b + r of the 64-bit value V = w1:w0 sits at byte
// b of V >> 2 r, so v_qsad_pk_u16_u8 (the sum of absolute byte differences of the four byte-aligned windows of a 64-bit value
// against one 32-bit reference, four 16-bit sums) tests windows r, r + 4, r + 8, r + 12 in one instruction: 3 shifts + 4 SADs +
// 7 packed minima + 2 compares = 16 VALU instead of 47 (tools/ubench_wany.hip on the MI355X: 124 against 192 cycles of its
// SIMD per call -- the SADs issue at a quarter of the rate --, 0 differences over 10^9 random calls).
typedef unsigned short ihp_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
	return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(ihp_u16x2, a), __builtin_bit_cast(ihp_u16x2, b)));
}
__device__ __forceinline__ bool window_any(unsigned w0, unsigned w1, unsigned qh)
{
	const unsigned long long V = ((unsigned long long)w1 << 32) | w0;
	const unsigned long long s0 = __builtin_amdgcn_qsad_pk_u16_u8(V, qh, 0ull), s1 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 2, qh, 0ull);
	const unsigned long long s2 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 4, qh, 0ull), s3 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 6, qh, 0ull);
	const unsigned a = pk_min_u16((unsigned)s0, (unsigned)(s0 >> 32)), b = pk_min_u16((unsigned)s1, (unsigned)(s1 >> 32));
	const unsigned c = pk_min_u16((unsigned)s2, (unsigned)(s2 >> 32)), d = pk_min_u16((unsigned)s3, (unsigned)(s3 >> 32));
	const unsigned m = pk_min_u16(pk_min_u16(a, b), pk_min_u16(c, d));
	return (m & 0xffffu) == 0 || (m >> 16) == 0;
}
// The same for ONE wave-uniform pair (w0, w1): lane j tests window j; bit j of the result <-> window j equals qh.
__device__ __forceinline__ unsigned window_bits(unsigned w0, unsigned w1, unsigned qh)
{
	const int lane = lane_id();
	// (every lane compares, the mask picks the sixteen: `lane < 16 && ...` is a branch around the compare and a round trip of the
	// result through a register)
	return (unsigned)ballot(fsh(w1, w0, 2u * (unsigned)(lane & 15)) == qh) & 0xffffu;
}

// Are the n bases at bit offset xb of LDS dword array P (dword index xd) equal to those at (yd, yb)?  Lane l compares
// bases [16 l, 16 l + 16); n <= 1024.  Reads one dword past the last one needed on each side (slots are padded).
__device__ __forceinline__ bool bits_equal(const uint32_t *P, int xd, unsigned xb, int yd, unsigned yb, int n)
{
	const int lane = lane_id();
	// every lane loads (the lanes past the end load the last pair again) and the mask decides: no branch around the loads
	const int last = (n - 1) >> 4, l = lane < last ? lane : last;
	const unsigned a0 = P[xd + l], a1 = P[xd + l + 1], b0 = P[yd + l], b1 = P[yd + l + 1];
	unsigned x = fsh(a1, a0, xb) ^ fsh(b1, b0, yb);
	const int rem = n - 16 * lane;
	x = rem >= 16 ? x : rem <= 0 ? 0u : x & ((1u << (2 * rem)) - 1u);
	return ballot(x != 0) == 0;
}

// Copy nbits bits from (sd, sbit) to (dd, dbit) inside P (bit offsets may exceed 31; they are relative to the dword
// indices sd / dd).  Lanes take destination dwords; bits of a destination dword outside the range are kept.  Source
// and destination must not overlap.  Reads P[sd - 1 ..] at most one dword outside the source range (padded).
// FRESH_LANE: the lane number is taken afresh in every call (k_asm_combine3: the compiler otherwise computes -32 * lane once per
// kernel, finds no register for it across a region and reloads it from scratch memory -- a wait -- in front of every copy).
template <bool FRESH_LANE = false>
__device__ __forceinline__ void copy_bits(uint32_t *P, int dd, int dbit, int sd, int sbit, int nbits)
{
	if (nbits <= 0) return;
	int lane = lane_id();
	if (FRESH_LANE) asm volatile("" : "+v"(lane));
	const int s = sbit - dbit;                                  // source bit of destination bit b is b + s
	const int d_first = dbit >> 5, d_last = (dbit + nbits - 1) >> 5;
	for (int d = d_first + lane; d <= d_last; d += 64) {
		const int sb = 32 * d + s;                              // source bit of this dword's bit 0 (may be negative by < 32)
		const int si = sb >> 5;                                 // arithmetic shift: floor
		const unsigned v = fsh(P[sd + si + 1], P[sd + si], (unsigned)sb & 31u);
		const int blo = dbit - 32 * d, bhi = dbit + nbits - 32 * d;      // valid bits of this dword: [blo, bhi)
		unsigned mask = 0xffffffffu;
		if (blo > 0) mask &= 0xffffffffu << blo;
		if (bhi < 32) mask &= (1u << bhi) - 1u;
		P[dd + d] = (P[dd + d] & ~mask) | (v & mask);
	}
}


// lanes [lo, hi] of the wave, 0 <= lo, hi <= 63; empty when hi < lo
__device__ __forceinline__ unsigned long long lane_range64(int lo, int hi)
{
	return hi < lo ? 0ull : ((~0ull << lo) & (~0ull >> (63 - hi)));
}

// ------------------------------------------------------------------------------------------------ read phase
constexpr int V2_MIN_READ = 20, V2_MAX_READ = 960;       // trimmed read lengths this path takes
constexpr int V2_MAX_CONTIGS = 64;                       // one directory lane per contig
constexpr int V2_MAX_REGION_READS = 640;                 // = V3_MAXREADS_WIDE (asm3_dev.h)
constexpr int V2_WL = 128;                               // work-list entries in registers (two of them)
constexpr int V2_WLX = 192;                              // ... and in LDS behind them, for regions of long reads (contigs of many dwords)
constexpr int V2_HDR = 8, V2_DIRW = 8;                   // hand-over record: header dwords, dwords per contig
constexpr int V2_HB_DW = 32;                             // dwords of the contig-head bit map (1024 bits)
__device__ __forceinline__ unsigned v2_head_hash(unsigned w) { return w ^ (w >> 10) ^ (w >> 20); }      // (masked to the map's size by the caller)

__device__ __forceinline__ int wl_make(int dword, int c, int k) { return (dword << 2) | (c << 16) | (k << 22); }
__device__ __forceinline__ unsigned dpp_wave_shl1(unsigned v)
{   // lane l gets v[l + 1]; lane 63 gets 0
	return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ long long bcast64(long long v, int src)
{
	return ((long long)bcast((int)(v >> 32), src) << 32) | (unsigned)bcast((int)v, src);
}

// Cost class of a region for the combine kernel (0 = most work): combine compares every contig with a growing list
// (contig.nim:254-281), so its time grows with the square of the contigs the read phase leaves.  k_asm_reads files every
// region under its class and k_asm_combine3 takes the classes in order -- the longest chains start first and the short ones
// fill the end of the launch instead of the other way round.
constexpr int LPT_CLASSES = 16, LPT_TIERS = 4;     // three combine launches: arenas of growing size at falling occupancy; a fourth for the regions of more than 255 reads (16-bit supports)
// (16 classes: two contigs apart over the range of 150 bp pile-ups, wider above.  The contig count is the best predictor the
// read phase has -- correlation with the measured cycles 0.88 on C2, 0.65 on C5; counting the multi-read contigs in made it worse)
__device__ __forceinline__ int lpt_class(int n)
{
	return n >= 48 ? 0 : n >= 40 ? 1 : n >= 34 ? 2 : n >= 29 ? 3 : n >= 25 ? 4 : n >= 22 ? 5 : n >= 19 ? 6 : n >= 17 ? 7 : n >= 15 ? 8
	     : n >= 13 ? 9 : n >= 11 ? 10 : n >= 9 ? 11 : n >= 7 ? 12 : n >= 5 ? 13 : n >= 3 ? 14 : 15;
}

// Candidate ranking of best_match (contig.nim:32-36, :107, :239) for exact matches: more matches, then the earlier contig,
// then the target-offset phase before the query-offset phase, then the smaller offset.
// One key carries the whole order: matches (10 bits: a read is at most V2_MAX_READ long) | 63 - contig | 1 - phase | 8191 - offset
// (a contig is at most MAXLEN = 8192 long); a larger key beats a smaller one, 0 = nothing found yet.
struct Best2 { int found, ma, c, ph, o; unsigned key; };
__device__ __forceinline__ unsigned best_key(int cn, int c, int ph, int o)
{
	return (unsigned)cn << 20 | (unsigned)(63 - c) << 14 | (unsigned)(1 - ph) << 13 | (unsigned)(8191 - o);
}
__device__ __forceinline__ bool beats(const Best2 &b, int cn, int c, int ph, int o) { return best_key(cn, c, ph, o) > b.key; }

// What the read kernel needs of a batch (a small struct: the ~70 fields of AsmArgs do not fit the scalar registers and
// were reloaded from spill lanes all through the read loop).
struct ReadArgs {
	const long long *region_read_off, *read_off, *read_start;
	const uint8_t *mapq, *read_skip, *v2_read_bad;
	const int *v2_trim_lo, *v2_trim_hi;
	const uint32_t *v2_pk;
	uint32_t *v2_hand; const long long *v2_hoff;
	int *n_final;                                             // per region: 0 for a region handed to out_list (nobody may take that list in this run)
	int *lpt_cnt, *lpt_seg; int lpt_stride;                   // regions for k_asm_combine3 by arena tier and cost class, longest first (see lpt_class)
	int *n_tier_b;                                            // counts the regions filed under the second and third tier (diagnostics)
	int tier_a_cap, tier_b_cap;                               // arena capacities of the first two combine launches: a region is filed under the first that holds it
	int tier_a_maxc;                                          // ... and has no more contigs than the first launch keeps a table for
	int manyc_thr; int *n_manyc;                              // counts the regions with more than manyc_thr contigs (the short table's size): what decides the first tier's build
	int hist_cap[11]; int *hist;                               // hist[k] counts the regions whose need fits hist_cap[k] (ascending) and no smaller one: what the
	                                                          // next batch of this shape sizes its first tier from (TierHint, indelope_hip.hip)
	double min_overlap_pct;
	int min_mapq_assemble, v2_pdw, n_regions;
	const int *in_list, *n_in; int *out_list, *n_out; int *work_counter;
	long long *prof; unsigned long long *t_start;
};

// Hand-over record of one region in HBM (a.v2_hand + a.v2_hoff[r], dwords), written by k_asm_reads, read by k_asm_combine3:
//   [0] contigs n (-1: the region was not taken), [1] reads of the region, [2..7] unused
//   n x V2_DIRW: { packed data offset (dwords from the record's start), length, nreads, start lo, start hi, anchor, 0, 0 }
//   one record dword per read of the region (see below), then the contigs' packed bases (ceil(len / 16) dwords each)
// Capacity (host, v2_hoff): 8 + 9 min(64, reads) + reads + bases / 16 + 8 dwords.
//
// Packed area P (LDS dwords) of one wave:
//   [0]                   zero pad (copy_bits may read the dword before a source)
//   [1, 1 + QW)           the read being inserted, from bit 0 (QW = dwords of the longest read + 2 of zero padding)
//   [RECB, RECB + reads)  one record per read of the region: contig (6 bits) | start relative to the contig's anchor,
//                         biased by 16384 (15 bits) | trimmed length (10 bits); 0xffffffff = read not inserted
//   [WLXB, WLXB + V2_WLX) work-list entries beyond the 128 in registers (regions with reads longer than 200 bases only)
//   [SLOT0, p_dwords)     contig slots, bump allocated, every slot followed by a pad dword
// The supports are not touched here: a read adds 1 to every base it covers (contig.nim:198-200, :216-219 with
// q.support == 1 and no corrections), so the support of a base is the number of records that cover it; positions are
// kept relative to an anchor that moves when bases are prepended (contig.nim:180-205).
__device__ inline int v2_read_phase(const ReadArgs &a, uint32_t *P, int p_dwords, int r, long long *prof, int &n_contigs, int &arena_need)
{
	const int lane = lane_id();
	const long long r0 = uni(a.region_read_off[r]), r1 = uni(a.region_read_off[r + 1]);
	const int nrr = (int)(r1 - r0);
	if (nrr > V2_MAX_REGION_READS) return IHP_E_CAPACITY;           // (what the wide combine build keeps records for; gen_roi caps a roi at 600 reads, indelope.nim:515)
	long long tp_ = prof ? (long long)clock64() : 0;             // diagnostics: cycles per stage into prof[12..15] (prep, target filter, query phase, insert); [7] set-up
#define V2_LAP(k) do { if (prof) { const long long t_ = (long long)clock64(); if (lane == 0) prof[k] += t_ - tp_; tp_ = t_; } } while (0)
	// ---- pass 0: which reads take part (indelope.nim:164-165), preconditions, shortest / longest trimmed read
	int minlen = 0x7fffffff, maxlen = 0;
	bool viol = false;
	for (long long g0 = r0; g0 < r1; g0 += 64) {
		const long long my = g0 + lane;
		if (my < r1 && a.mapq[my] >= a.min_mapq_assemble && !(a.read_skip && a.read_skip[my])) {
			const int tl = a.v2_trim_hi[my] - a.v2_trim_lo[my];
			viol |= a.v2_read_bad[my] != 0 || tl < V2_MIN_READ || tl > V2_MAX_READ;
			minlen = tl < minlen ? tl : minlen; maxlen = tl > maxlen ? tl : maxlen;
		}
	}
	if (ballot(viol)) return IHP_E_CAPACITY;
	minlen = wave_min_i32(minlen); maxlen = wave_max_i32s(maxlen);
	const int mo_min = maxlen ? (int)(a.min_overlap_pct * (double)minlen) : 17;
	if (mo_min < 17) return IHP_E_CAPACITY;                  // a 16-base window must lie inside every acceptable overlap
	if (maxlen - (int)(a.min_overlap_pct * (double)maxlen) > 127) return IHP_E_CAPACITY;   // query offsets 1..127 (two registers)
	const int QW = ((maxlen + 15) >> 4) + 2;
	const int RECB = 1 + QW, WLXB = RECB + nrr;
	const int wlx_n = maxlen > 200 ? V2_WLX : 0;              // work-list entries beyond the two registers
	// Round 6, a 1024-bit map of the contigs' HEADS (the hash of a contig's first 16 bases; bits are only ever set): a query offset
	// (contig.nim:114-135) needs a window of the read that IS some contig's head, which in BAM order almost never exists -- one
	// lookup per window lane says so instead of a walk over the offsets or the contigs (a set bit only costs the walk it used to).
	// (long reads: up to 127 query offsets a read and more contigs -- four times the bits)
	const int hb_dw = maxlen > 200 ? 4 * V2_HB_DW : V2_HB_DW;
	const unsigned hb_mask = 32u * (unsigned)hb_dw - 1u;
	const int HBB = WLXB + wlx_n, SLOT0 = HBB + hb_dw;
	if (SLOT0 + 8 > p_dwords) return IHP_E_CAPACITY;
	int bump = SLOT0;
	if (lane == 0) P[0] = 0;
	for (int i = lane; i < hb_dw; i += 64) P[HBB + i] = 0;
	auto head_seen = [&](unsigned head) {                        // wave-uniform head: one lane sets its bit
		const unsigned h = v2_head_hash(head) & hb_mask;
		if (lane == 0) P[HBB + (int)(h >> 5)] |= 1u << (h & 31u);
	};
	const long long base_idx = nrr ? (uni(a.read_off[r0]) >> 4) + r0 : 0;   // the region's first packed dword
	const uint32_t *pkp = a.v2_pk + base_idx;
	// contig directory: lane c <-> contig c (creation order = list order of contig.nim:243-248)
	int d_woff = 0, d_len = 0, d_capw = 0, d_nreads = 0, d_slo = 0, d_shi = 0, d_anchor = 0;
	unsigned d_head = 0;
	int wl0 = 0, wl1 = 0, wl_n = 0;                           // work list: lane e of wl0 / wl1 = entry e / 64 + e
	int n = 0;
	auto wl_append = [&](int dword, int c, int k) -> bool {
		if (wl_n >= V2_WL + wlx_n) return false;
		const int e = wl_make(dword, c, k);
		if (wl_n < 64) wl0 = lane == wl_n ? e : wl0;
		else if (wl_n < V2_WL) wl1 = lane == wl_n - 64 ? e : wl1;
		else { if (lane == 0) P[WLXB + wl_n - V2_WL] = (uint32_t)e; LDS_ORDER(); }
		wl_n++;
		return true;
	};
	auto n_entries = [&](int len) { return len >= mo_min ? (len - mo_min) / 16 + 1 : 0; };   // dwords that hold a valid target offset

	for (long long g0 = r0; g0 < r1; g0 += 64) {
		const long long my = g0 + lane;
		long long mstart = 0; int midx = 0, mpack = 0; bool ok = false;
		if (my < r1) {
			const long long moff = a.read_off[my];
			mstart = a.read_start[my];
			const int tlo = a.v2_trim_lo[my], tl = a.v2_trim_hi[my] - tlo;
			ok = a.mapq[my] >= a.min_mapq_assemble && !(a.read_skip && a.read_skip[my]);
			mstart += tlo;                                       // read.start + o (indelope.nim:169)
			midx = (int)((moff >> 4) + my + (tlo >> 4) - base_idx);
			const int mo = ok ? (int)(a.min_overlap_pct * (double)tl) : 0;      // :169, for 64 reads at once
			mpack = (tl & 1023) | ((tlo & 15) << 10) | (mo << 14);
		}
		unsigned long long elig = ballot(ok), topf = elig;       // topf: reads whose packed bases have not been requested yet
		int rec = -1;
		// the packed bases of the next reads are in flight while the current one is matched
		unsigned raw0 = 0, raw1 = 0, raw2 = 0, raw3 = 0;
		auto fetch = [&]() -> unsigned {
			if (!topf) return 0u;
			const int k = ctz64(topf);
			topf &= topf - 1;
			const int pk = bcast(mpack, k);
			const int nq = (((pk >> 10) & 15) + (pk & 1023) + 15) >> 4;     // dwords that hold the trimmed read (before shifting)
			return lane <= nq ? pkp[bcast(midx, k) + lane] : 0u;             // one more than needed: the funnel shift's upper half
		};
		raw0 = fetch(); raw1 = fetch(); raw2 = fetch(); raw3 = fetch();
		V2_LAP(7);
		while (elig) {
			const int k = ctz64(elig);
			elig &= elig - 1;
			const int pk = bcast(mpack, k);
			const int tl = pk & 1023, mo = (unsigned)pk >> 14;
			const int omin = tl - mo;                            // contig.nim:78 (abs: tl >= mo because pct <= 1)
			// ---- the read: bits from 0, zero beyond its end; lane l holds bases [16 l, 16 l + 16)
			unsigned rq = fsh(dpp_wave_shl1(raw0), raw0, 2u * (unsigned)((pk >> 10) & 15));
			{
				const int rem = tl - 16 * lane;
				rq = rem <= 0 ? 0u : rem < 16 ? rq & ((1u << (2 * rem)) - 1u) : rq;
			}
			raw0 = raw1; raw1 = raw2; raw2 = raw3; raw3 = fetch();
			if (lane < QW) P[1 + lane] = rq;                     // QW <= 62
			LDS_ORDER();
			const unsigned qh = (unsigned)__builtin_amdgcn_readlane((int)rq, 0);
			// windows of the read at offsets lane and 64 + lane (query-offset phase)
			unsigned wq0, wq1 = 0;
			{
				const int dw = 1 + (lane >> 4);
				wq0 = fsh(P[dw + 1], P[dw], 2u * (unsigned)(lane & 15));
				if (omin > 63) { const int d2 = dw + 4; wq1 = fsh(d2 + 1 < RECB ? P[d2 + 1] : 0u, d2 < RECB ? P[d2] : 0u, 2u * (unsigned)(lane & 15)); }
			}
			Best2 best = {0, 0, 0, 0, 0, 0u};
			V2_LAP(12);
			// ---- target offsets (contig.nim:81-111): one dword of one contig per lane
			// (the two register chunks written out, the LDS chunks of long-read regions in a loop behind them: choosing the chunk's
			// source inside one loop cost a dozen scalar instructions and three branches per chunk)
			auto chunk = [&](const int h, const int ent) {
				unsigned w0 = 0, w1 = 0;
				bool any = false;
				if (h * 64 + lane < wl_n) {
					const int dw = (ent & 0xffff) >> 2;
					w0 = P[dw]; w1 = P[dw + 1];
					any = window_any(w0, w1, qh);
				}
				unsigned long long hm = ballot(any);
				while (hm) {
					const int e = ctz64(hm);
					hm &= hm - 1;
					unsigned bits = window_bits((unsigned)__builtin_amdgcn_readlane((int)w0, e), (unsigned)__builtin_amdgcn_readlane((int)w1, e), qh);
					const int en = __builtin_amdgcn_readlane(ent, e);
					const int c = (en >> 16) & 63, kk = (unsigned)en >> 22;
					const int tlen = bcast(d_len, c), woff = bcast(d_woff, c);
					while (bits) {
						const int o = 16 * kk + __builtin_ctz(bits);
						bits &= bits - 1;
						if (o > tlen - mo) continue;                 // :79 offsets 0 .. len(t) - min_overlap
						const int cn = tl < tlen - o ? tl : tlen - o;
						if (!beats(best, cn, c, 0, o)) continue;
						if (bits_equal(P, 1, 0u, woff + (o >> 4), 2u * (unsigned)(o & 15), cn)) { best.found = 1; best.ma = cn; best.c = c; best.ph = 0; best.o = o; best.key = best_key(cn, c, 0, o); }
					}
				}
			};
			if (wl_n > 0) chunk(0, wl0);
			if (wl_n > 64) chunk(1, wl1);
			for (int h = 2; h * 64 < wl_n; ++h) chunk(h, h * 64 + lane < wl_n ? (int)P[WLXB + (h - 2) * 64 + lane] : 0);
			V2_LAP(13);
			// ---- query offsets 1 .. omin (contig.nim:114-135): lane o holds the read's window, the contigs' heads come by.
			// Hits are rare (the read would have to extend a contig to the left): first only whether there is one at all.
			// A query offset o matches at most tl - o bases, so once the target offsets have found `best.ma` matches only
			// o <= tl - best.ma can still win (more matches, or as many on an earlier contig) -- for a read that extends its contig
			// by a few bases that is a few offsets, and walking THEM (the window of offset o against every contig's head at once,
			// lane <-> contig) costs a handful of compares where walking the contigs costs one per contig.
			int kq = best.found ? (omin < tl - best.ma ? omin : tl - best.ma) : omin;
			if (kq >= 1) {
				// is any window at an offset 1 .. kq the head of some contig?  (the bit map says "no" exactly; "maybe" takes the walk)
				const unsigned h0 = v2_head_hash(wq0) & hb_mask;
				bool maybe = lane >= 1 && lane <= kq && ((P[HBB + (int)(h0 >> 5)] >> (h0 & 31u)) & 1u);
				if (kq > 63) { const unsigned h1 = v2_head_hash(wq1) & hb_mask; maybe |= 64 + lane <= kq && ((P[HBB + (int)(h1 >> 5)] >> (h1 & 31u)) & 1u); }
				if (!ballot(maybe)) kq = 0;
			}
			if (kq >= 1 && kq <= 12) {
				const unsigned long long live = n >= 64 ? ~0ull : (1ull << n) - 1ull;
				for (int o = 1; o <= kq; ++o) {
					const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wq0, o);
					unsigned long long m = ballot(d_head == w) & live;
					while (m) {
						const int c = ctz64(m);
						m &= m - 1;
						const int tlen = bcast(d_len, c), woff = bcast(d_woff, c);
						const int cn = tl - o < tlen ? tl - o : tlen;
						if (cn < mo - 1 || !beats(best, cn, c, 1, o)) continue;      // best_ma starts at min_overlap - 1 (:81, :107)
						if (bits_equal(P, 1 + (o >> 4), 2u * (unsigned)(o & 15), woff, 0u, cn)) { best.found = 1; best.ma = cn; best.c = c; best.ph = 1; best.o = o; best.key = best_key(cn, c, 1, o); }
					}
				}
			} else if (kq >= 1) {
				const unsigned long long v0 = lane_range64(1, kq < 63 ? kq : 63), v1 = kq > 63 ? lane_range64(0, kq - 64) : 0ull;
				unsigned long long any0 = 0, any1 = 0;
				for (int c = 0; c < n; c += 4) {                     // lanes >= n hold head 0: a false hit only costs the second look
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const unsigned hd = (unsigned)__builtin_amdgcn_readlane((int)d_head, c + u);
						any0 |= ballot(wq0 == hd);
						if (kq > 63) any1 |= ballot(wq1 == hd);
					}
				}
				if ((any0 & v0) | (any1 & v1)) {
					for (int c = 0; c < n; ++c) {
						const unsigned hd = (unsigned)__builtin_amdgcn_readlane((int)d_head, c);
						const unsigned long long m0 = ballot(wq0 == hd) & v0, m1 = kq > 63 ? ballot(wq1 == hd) & v1 : 0ull;
						if (!(m0 | m1)) continue;
						const int tlen = bcast(d_len, c), woff = bcast(d_woff, c);
						for (int half = 0; half < 2; ++half) {
							unsigned long long m = half ? m1 : m0;
							while (m) {
								const int o = 64 * half + ctz64(m);
								m &= m - 1;
								const int cn = tl - o < tlen ? tl - o : tlen;
								if (cn < mo - 1 || !beats(best, cn, c, 1, o)) continue;      // best_ma starts at min_overlap - 1 (:81, :107)
								if (bits_equal(P, 1 + (o >> 4), 2u * (unsigned)(o & 15), woff, 0u, cn)) { best.found = 1; best.ma = cn; best.c = c; best.ph = 1; best.o = o; best.key = best_key(cn, c, 1, o); }
							}
						}
					}
				}
			}
			V2_LAP(14);
			// ---- insert (contig.nim:243-248)
			int rec_c, rec_s;
			if (best.found) {
				const int c = best.c, off = best.ph ? -best.o : best.o, aoff = best.o;
				const int tlen = bcast(d_len, c), woff = bcast(d_woff, c), capw = bcast(d_capw, c), anchor = bcast(d_anchor, c);
				int newlen;
				if (off < 0) { newlen = aoff + tlen; if (tl > newlen) newlen = tl; }          // contig.nim:180-195
				else { newlen = tlen; if (off + tl > newlen) newlen = off + tl; }               // :210-213
				if (newlen > MAXLEN) return IHP_E_CAPACITY;
				const bool sel = lane == c;
				if (off >= 0 && newlen <= 16 * capw) {
					copy_bits(P, woff, 2 * tlen, 1, 2 * (tlen - off), 2 * (newlen - tlen));      // :220-221 new bases
					d_len = sel ? newlen : d_len;
					rec_s = off - anchor;
				} else {
					const int ncapw = (newlen + headroom(newlen) + 15) >> 4;
					if (bump + ncapw + 1 > p_dwords) return IHP_E_CAPACITY;
					const int nw = bump;
					for (int i = lane; i <= ncapw; i += 64) P[nw + i] = 0;
					LDS_ORDER();
					if (off < 0) {
						copy_bits(P, nw, 0, 1, 0, 2 * aoff);                                       // :184-185
						LDS_ORDER();
						copy_bits(P, nw, 2 * aoff, woff, 0, 2 * tlen);                             // :187-188
						LDS_ORDER();
						copy_bits(P, nw, 2 * (aoff + tlen), 1, 2 * (aoff + tlen), 2 * (tl - aoff - tlen));   // :191-195
					} else {
						copy_bits(P, nw, 0, woff, 0, 2 * tlen);
						LDS_ORDER();
						copy_bits(P, nw, 2 * tlen, 1, 2 * (tlen - off), 2 * (newlen - tlen));
					}
					LDS_ORDER();
					bump += ncapw + 1;
					const unsigned nh = (unsigned)uni((int)P[nw]);
					d_woff = sel ? nw : d_woff; d_capw = sel ? ncapw : d_capw; d_len = sel ? newlen : d_len;
					d_head = sel ? nh : d_head;
					if (off < 0) head_seen(nh);                              // (a prepend: the contig has a new head)
					if (off < 0) {
						const long long qstart = bcast64(mstart, k);
						d_slo = sel ? (int)qstart : d_slo; d_shi = sel ? (int)(qstart >> 32) : d_shi;   // t.start = q.start (:204)
						d_anchor = sel ? anchor + aoff : d_anchor;
					}
					rec_s = off < 0 ? -(anchor + aoff) : off - anchor;
					// the contig's work-list entries point into the new slot
					{
						const bool m0 = ((wl0 >> 16) & 63) == c && lane < wl_n, m1 = ((wl1 >> 16) & 63) == c && 64 + lane < wl_n;
						wl0 = m0 ? wl_make(nw + (int)((unsigned)wl0 >> 22), c, (int)((unsigned)wl0 >> 22)) : wl0;
						wl1 = m1 ? wl_make(nw + (int)((unsigned)wl1 >> 22), c, (int)((unsigned)wl1 >> 22)) : wl1;
						for (int i = lane; i < wl_n - V2_WL; i += 64) {
							const int e = (int)P[WLXB + i];
							if (((e >> 16) & 63) == c) P[WLXB + i] = (uint32_t)wl_make(nw + (int)((unsigned)e >> 22), c, (int)((unsigned)e >> 22));
						}
						LDS_ORDER();
					}
				}
				d_nreads = sel ? d_nreads + 1 : d_nreads;                                       // :203, :222
				{
					const int w2 = bcast(d_woff, c);
					for (int kk = n_entries(tlen); kk < n_entries(newlen); ++kk) if (!wl_append(w2 + kk, c, kk)) return IHP_E_CAPACITY;
				}
				rec_c = c;
			} else {                                                                            // contigs.add(q) (:248)
				if (n >= V2_MAX_CONTIGS) return IHP_E_CAPACITY;
				const int capw = (tl + 32 + 15) >> 4;
				if (bump + capw + 1 > p_dwords) return IHP_E_CAPACITY;
				const int nw = bump;
				for (int i = lane; i <= capw; i += 64) P[nw + i] = i < QW ? rq : 0u;             // lane i holds dword i of the read (QW <= 62)
				bump += capw + 1;
				const bool sel = lane == n;
				const long long qstart = bcast64(mstart, k);
				d_woff = sel ? nw : d_woff; d_capw = sel ? capw : d_capw; d_len = sel ? tl : d_len; d_head = sel ? qh : d_head;
				head_seen(qh);
				d_nreads = sel ? 1 : d_nreads; d_slo = sel ? (int)qstart : d_slo; d_shi = sel ? (int)(qstart >> 32) : d_shi;
				d_anchor = sel ? 0 : d_anchor;
				for (int kk = 0; kk < n_entries(tl); ++kk) if (!wl_append(nw + kk, n, kk)) return IHP_E_CAPACITY;
				rec_c = n; rec_s = 0;
				n++;
			}
			LDS_ORDER();
			rec = lane == k ? (rec_c | ((rec_s + 16384) << 6) | (tl << 21)) : rec;
			V2_LAP(15);
		}
		if (my < r1) P[RECB + (int)(my - r0)] = (uint32_t)rec;
	}
	LDS_ORDER();
	// ---- hand-over record in HBM: header, directory, records, packed bases
	uint32_t *H = a.v2_hand + uni(a.v2_hoff[r]);
	const int nd = lane < n ? (d_len + 15) >> 4 : 0;
	const unsigned pincl = wave_scan_add((unsigned)nd);
	const int pbase = V2_HDR + V2_DIRW * n + nrr;
	const int poff = pbase + (int)pincl - nd;
	if (lane == 0) { H[0] = (uint32_t)n; H[1] = (uint32_t)nrr; }
	if (lane < n) {
		uint4 a0, a1;
		a0.x = (uint32_t)poff; a0.y = (uint32_t)d_len; a0.z = (uint32_t)d_nreads; a0.w = (uint32_t)d_slo;
		a1.x = (uint32_t)d_shi; a1.y = (uint32_t)d_anchor; a1.z = 0; a1.w = 0;
		*(uint4 *)(H + V2_HDR + V2_DIRW * lane) = a0;
		*(uint4 *)(H + V2_HDR + V2_DIRW * lane + 4) = a1;
	}
	for (int i = lane; i < nrr; i += 64) H[V2_HDR + V2_DIRW * n + i] = P[RECB + i];
	for (int c = 0; c < n; ++c) {
		const int w = bcast(d_woff, c), cnt = bcast(nd, c), po = bcast(poff, c);
		for (int i = lane; i < cnt; i += 64) H[po + i] = P[w + i];
	}
	V2_LAP(7);
#undef V2_LAP
	n_contigs = n;
	// what combine (asm3_dev.h) needs of its two areas.
	// In capacity units C of a combine launch (C bytes of supports, C / 8 + 128 dwords of packed bases): support bytes are kept
	// for multi-read contigs only (half of the single-read ones is set aside for those that will merge), a merge takes a
	// slot of its new length in both areas, nothing is compacted until room runs out.
	{
		const bool multi = lane < n && d_nreads != 1;
		const int bl = lane < n ? align4(d_len) + SLOT_PAD : 0;
		const int mxl = wave_max_i32s(lane < n ? d_len : 0);
		const int need_sup = wave_sum_i(multi ? bl : bl >> 1) + 2 * mxl + 256;
		const int need_pm = wave_sum_i(lane < n ? ((d_len + 15) >> 4) + 1 : 0) + ((2 * mxl + 15) >> 4) + 8;
		const int c_pm = 8 * (need_pm - 128);
		arena_need = need_sup > c_pm ? need_sup : c_pm;
	}
	return 0;
}

__device__ __forceinline__ bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

}  // namespace ihp
