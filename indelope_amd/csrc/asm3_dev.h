// asm3_dev.h -- combine (contig.nim:254-281) on what the read phase leaves: 2-bit packed bases and u8 supports, both in LDS.
//
// Round 2's k_asm_combine unpacked the hand-over record into a byte arena (LDS), kept the supports in HBM (4 B per base;
// every trim, merge and min/max update was an HBM round trip in the region's serial chain: 57% of its wave cycles were
// s_waitcnt) and mirrored the bytes in packed form for the exact scans.  Here the packed bases are the only copy and the
// supports are one byte per base beside them (a region of at most 255 reads cannot exceed 255 on any base: a base's
// support is a number of distinct reads), 1.25 B per base as before -- and nothing of a region's chain leaves the CU
// until the epilogue writes the final contigs.
//   * exact candidates of a best_match call (16-base seed hits of the target- and query-offset phases) are collected into
//     lanes and verified TOGETHER: lane <-> (candidate, 16-base chunk), one LDS round trip for all of them instead of one
//     dependent round trip per seed hit; the winner is the smallest (matches desc, contig asc, phase, offset asc) key;
//   * pairs for which the vote rule (contig.nim:44-47) can fire are scanned with lanes <-> offsets on the packed bases;
//     the supports are only looked at where two bases differ;
//   * corrections live in LDS; insert (contig.nim:156-222) is funnel-shift copies of packed dwords and byte adds.
// Preconditions (anything else is handed to the byte-based passes, whose results are identical): the region has at most
// 256 reads and no base is covered by all 256 of them, max_mismatch == 0, contigs shorter than 2048 bases, at most 64 contigs, combine_min_overlap >= 17.
// Round 6, the WIDE build (V3StateT<MAXC, true>): the same code with 16-bit supports and room for the records of 640 reads, for
// the regions the reference admits and the byte build cannot hold -- gen_roi hands over up to 600 reads per roi
// (indelope.nim:483-485, :515), and a pile-up that deep has bases with more than 256 reads on them.  Offsets into the support
// area (so, bump_sup, sup_cap) are in ELEMENTS in both builds; only the element type differs.
#pragma once
#include <type_traits>
#include "asm2_dev.h"

namespace ihp {

constexpr int V3_MAXC = 64;
constexpr int V3_MAXLEN = 2047;           // longest contig: 11 bits in the ranking key
constexpr int V3_CORR = 1024;             // most corrections one merge may have (they sit at the top of the packed area, counting down)
constexpr int V3_NOZONE = 0x3fff;

// (MAXC: the first tier's launch keeps room for 32 contigs -- what nearly every region has when the read phase is done --,
// 1.7 KB instead of 3.2: LDS is what decides how many regions a CU holds, and a region's chain is latency bound, so the launch
// runs as fast as it has regions resident: 16 per CU 2.83 ms per 100 000 C2 regions, 14: 3.14, 12: 3.95, 10: 4.74.)
constexpr int V3_MAXREADS_WIDE = 640;     // reads of a region the wide build takes (ten record registers)
template <int MAXC_, bool WIDE_ = false>
struct V3StateT {                         // static LDS, one per wave
	static constexpr int MAXC = MAXC_;
	static constexpr bool WIDE = WIDE_;
	typedef typename std::conditional<WIDE_, unsigned short, unsigned char>::type sup_t;   // one support
	static constexpr unsigned SUP_MAX = WIDE_ ? 65535u : 255u;
	static constexpr int NREC = WIDE_ ? V3_MAXREADS_WIDE / 64 : 4;     // record registers of the take-over (64 reads each)
	static constexpr int CORR_DW = WIDE_ ? 2 : 1;                       // dwords per correction (the wide build keeps the site's final support in a dword of its own)
	int dw[MAXC_];                      // packed slot: first dword in PM
	int so[MAXC_];                      // support slot: first element in SUP; -1: a contig of one read, support 1 on every base, nothing kept
	int len[MAXC_], cap[MAXC_];       // bases; cap = bases both slots have room for from the current start
	int nreads[MAXC_];
	long long start[MAXC_];
	short lo3[MAXC_], hi3[MAXC_];     // every base in [lo3, hi3) has support >= 3 and no other has (V3_NOZONE / 0: not one run)
	short loT[MAXC_], hiT[MAXC_];     // the longest run of bases with support >= v3_thr(nreads): bases no vote can overrule (see v3_slide_votes)
	unsigned char sh[MAXC_];            // bases into dword dw where the contig starts (trim moves it)
	sup_t smin[MAXC_], smax[MAXC_];
	short listA[MAXC_], listB[MAXC_];
	short qt[MAXC_], mt[MAXC_];       // step at which the contig was the query of pass 1 (0: never), step of its last change (0: none)
	long long prof[16];
	int cnt[16];                          // diagnostics (profile): see ihp_batch_profile [32..47]
};

struct V3Ctx {
	uint32_t *PM; uint8_t *SUP;           // SUP first, PM right behind it (one dynamic LDS block)
	int pm_cap, sup_cap;                  // dwords, supports (bytes, or 16-bit words in the wide build: sup_ld / sup_st)
	int bump_pm, bump_sup;
	unsigned long long alive;             // contigs (slots of V3State) whose bases and supports are still needed
	int clock;                            // best_match calls of this region so far, over both passes (see V3State::qt)
	long long *prof; int *cnt;
};
#define V3_CNT(C, k, n) do { if ((C).cnt && lane_id() == 0) (C).cnt[k] += (n); } while (0)

template <class ST> __device__ __forceinline__ unsigned sup_ld(const V3Ctx &C, int i) { return ((const typename ST::sup_t *)C.SUP)[i]; }
template <class ST> __device__ __forceinline__ void sup_st(const V3Ctx &C, int i, unsigned v) { ((typename ST::sup_t *)C.SUP)[i] = (typename ST::sup_t)v; }

#define V3_T0(C) const long long t0_ = (C).prof ? (long long)clock64() : 0
#define V3_T1(C, k) do { if ((C).prof && lane_id() == 0) (C).prof[k] += (long long)clock64() - t0_; } while (0)

// 16 bases from base address b (dword index * 16 + base in dword) of PM
__device__ __forceinline__ unsigned pk16(const uint32_t *PM, int b)
{
	const int d = b >> 4;
	return fsh(PM[d + 1], PM[d], 2u * (unsigned)(b & 15));
}
// differing bases of two 16-base words: bit 2k set <-> base k differs
__device__ __forceinline__ unsigned diff16(unsigned a, unsigned b) { const unsigned x = a ^ b; return (x | (x >> 1)) & 0x55555555u; }

// contig.nim:44-47 overrules a base only if its support s is below 3 AND the contig's reads exceed 3 s: a base with
// support >= min(3, ceil(nreads / 3)) can never be voted away.
__device__ __forceinline__ int v3_thr(int nreads) { return nreads >= 7 ? 3 : (nreads + 2) / 3; }

// Running summary of a contig's supports, 64 consecutive bases per step (every lane calls add(); `valid` = the lane holds a
// base): extrema, the ">= 3" bases when they form exactly one run (what the trim shortcut needs), and the LONGEST run of
// bases with support >= thr (what the vote filter needs: any 16 of them will do).
struct SupStats {
	unsigned mn, mx; int f3, l3, c3, thr;
	unsigned carry;                                      // 1 + the last base seen so far that is below thr (0: none)
	int blen, bend;                                      // per lane: the longest run ending at one of this lane's bases, and where
	__device__ __forceinline__ void init(int thr_) { mn = 65535u; mx = 0; f3 = 0x7fff; l3 = -1; c3 = 0; thr = thr_; carry = 0; blen = 0; bend = 0; }
	__device__ __forceinline__ void add(unsigned v, int i, bool valid)
	{
		if (valid) {
			mn = v < mn ? v : mn; mx = v > mx ? v : mx;
			if (v >= 3u) { f3 = i < f3 ? i : f3; l3 = i; c3++; }
		}
		const bool strong = valid && (int)v >= thr;
		unsigned pb = wave_scan_max(strong ? 0u : (unsigned)(i + 1));
		pb = pb > carry ? pb : carry;
		const int run = strong ? i + 1 - (int)pb : 0;                    // bases of the run that ends here
		if (run > blen) { blen = run; bend = i; }
		carry = (unsigned)__builtin_amdgcn_readlane((int)pb, 63);
	}
	template <class ST> __device__ __forceinline__ void store(ST &S, int c)
	{   // wave-uniform control flow; lane 0 writes
		mn = wave_min_u32(mn); mx = wave_max_u32(mx);
		f3 = wave_min_i32(f3); l3 = wave_max_i32s(l3); c3 = wave_sum_i(c3);
		const int best = wave_max_i32s(blen);
		const int end = best > 0 ? __builtin_amdgcn_readlane(bend, ctz64(ballot(blen == best))) : 0;
		if (lane_id() == 0) {
			S.smin[c] = (typename ST::sup_t)mn; S.smax[c] = (typename ST::sup_t)mx;
			const bool clean = l3 >= f3 && c3 == l3 - f3 + 1;
			S.lo3[c] = (short)(clean ? f3 : V3_NOZONE); S.hi3[c] = (short)(clean ? l3 + 1 : 0);
			S.loT[c] = (short)(best > 0 ? end + 1 - best : V3_NOZONE); S.hiT[c] = (short)(best > 0 ? end + 1 : 0);
		}
		LDS_ORDER();
	}
};

__device__ __forceinline__ bool allowed3(unsigned qs, unsigned ts, int qreads, int treads)
{   // contig.nim:44-47
	return (qs < 3u && ts > 3u * qs && qreads > 3 * (int)qs) || (ts < 3u && qs > 3u * ts && treads > 3 * (int)ts);
}

// ------------------------------------------------------------------------------------------------ the next work item
// The work queue of ihp_common.h (wq_next) taken apart.  A wave asks for its NEXT region beside the three rounds of loads the
// EPILOGUE of the current one makes anyway: the ticket (the shard's counter) beside the region's read range, the ticket's region
// (the cost classes' list) beside the reads' stops, that region's hand-over offset beside the reference offsets.  These were three
// dependent round trips to L2 in front of every take-over, queued behind the stores of the region before.  (Asked for at the
// START of a region -- beside the take-over's loads -- the ticket binds a region to a wave a whole region early: with two
// regions per wave, a batch of 10 000, that is a static schedule, and the launch's tail grew by a fifth.)  `stage` says how far
// the chain has come (a region without an epilogue leaves all of it to v3n_finish, one wait each).
typedef __attribute__((address_space(1))) int *v3_gint_p;
struct V3Next {
	v3_gint_p ctr; int S, s, n_items, cls_end;                     // the shard's counter; item = ticket * S + s; end of every cost class (lane c)
	int tick_v, rn_v; long long hn_v;                                // in flight (vector registers: the ticket in lane 0)
	int item, rn; long long hn;                                      // known: rn = -1: the queue is dry
	int stage;                                                       // 0 nothing asked, 1 ticket, 2 region, 3 offset in flight, 4 all known
};
__device__ __forceinline__ void v3n_ticket(V3Next &N)
{
	// (an address the compiler cannot call uniform: its atomic optimizer would put a v_readfirstlane -- a wait -- right behind the atomic)
	v3_gint_p p = N.ctr;
	asm volatile("" : "+v"(p));
	if (lane_id() == 0) N.tick_v = __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	N.stage = 1;
}
__device__ __forceinline__ void v3n_region(const AsmArgs &a, V3Next &N)
{
	// (every load of the chain is unconditional, ONE load from a chosen address, and its answer is first looked at in the NEXT
	// step: a value that meets a select or a join here would be waited for here)
	const int lane = lane_id();
	N.item = uni(N.tick_v) * N.S + N.s;
	const bool live = N.item < N.n_items;
	const int *p = (const int *)a.v2_hoff;                             // (no list: any address that can be read)
	if (a.lpt_cnt) {                                                   // item -> (class, position)
		const int c = popc64(ballot(lane < a.lpt_nclass && N.item >= N.cls_end));
		const int pos = N.item - (c ? __builtin_amdgcn_readlane(N.cls_end, c - 1) : 0);
		p = a.lpt_seg + (live ? (size_t)c * a.lpt_stride + pos : 0);
	} else if (a.in_list) p = a.in_list + (live ? N.item : 0);
	N.rn_v = *p;
	N.stage = 2;
}
__device__ __forceinline__ void v3n_offset(const AsmArgs &a, V3Next &N)
{
	N.rn = N.item < N.n_items ? (a.lpt_cnt || a.in_list ? uni(N.rn_v) : N.item) : -1;
	N.hn_v = a.v2_hoff[N.rn >= 0 ? N.rn : 0];
	N.stage = 3;
}
__device__ __forceinline__ void v3n_finish(const AsmArgs &a, V3Next &N)
{
	if (N.stage < 2) v3n_region(a, N);
	if (N.stage < 3) v3n_offset(a, N);
	if (N.stage < 4) { N.hn = uni(N.hn_v); N.stage = 4; }
}

// ------------------------------------------------------------------------------------------------ take-over
// Support extrema and zones of SUP[so .. so + len), one base per lane.
template <class ST>
__device__ inline void v3_stats(ST &S, const V3Ctx &C, int c)
{
	const int lane = lane_id();
	const int so = uni(S.so[c]), n = uni(S.len[c]);
	if (so < 0) {                                                    // a single read: 1 everywhere
		if (lane == 0) { S.smin[c] = 1; S.smax[c] = 1; S.lo3[c] = V3_NOZONE; S.hi3[c] = 0; S.loT[c] = 0; S.hiT[c] = (short)n; }
		LDS_ORDER();
		return;
	}
	SupStats st; st.init(v3_thr(uni(S.nreads[c])));
	for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; st.add(i < n ? sup_ld<ST>(C, so + i) : 0u, i, i < n); }
	st.store(S, c);
}

// Hand-over record of k_asm_reads -> directory in S, packed bases in PM, supports counted from the read records in SUP.
// Returns 1 if the read phase did not take the region, 0 when ready, IHP_E_CAPACITY when it does not fit / is not for this path.
template <class ST>
__device__ inline int v3_take_over(const AsmArgs &a, ST &S, V3Ctx &C, long long hoff, int &n_pre)
{
	const int lane = lane_id();
	const uint32_t *H = a.v2_hand + hoff;                             // (v2_hoff[r]: asked for in the epilogue of the region before this one)
	const int n = uni((int)H[0]), nrr = uni((int)H[1]);
	n_pre = 0;
	if (n < 0) return 1;
	n_pre = n;
	constexpr int NREC = ST::NREC, SB = (int)sizeof(typename ST::sup_t);
	if (nrr > 64 * NREC || n > ST::MAXC) return IHP_E_CAPACITY;     // (the records of a region are kept in four registers; ten in the wide build)
	int d_poff = 0, d_len = 0, d_nreads = 0, d_slo = 0, d_shi = 0, d_anchor = 0;
	if (lane < n) {
		const uint4 a0 = *(const uint4 *)(H + V2_HDR + V2_DIRW * lane), a1 = *(const uint4 *)(H + V2_HDR + V2_DIRW * lane + 4);
		d_poff = (int)a0.x; d_len = (int)a0.y; d_nreads = (int)a0.z; d_slo = (int)a0.w; d_shi = (int)a1.x; d_anchor = (int)a1.y;
	}
	const uint32_t *REC = H + V2_HDR + V2_DIRW * n;
	const int scap = lane < n && d_nreads != 1 ? align4(d_len) + SLOT_PAD : 0, pnd = lane < n ? ((d_len + 15) >> 4) + 1 : 0;
	const unsigned sincl = wave_scan_add((unsigned)scap), pincl = wave_scan_add((unsigned)pnd);
	const int soff = (int)sincl - scap, stotal = __builtin_amdgcn_readlane((int)sincl, 63);
	const int poff = (int)pincl - pnd, ptotal = __builtin_amdgcn_readlane((int)pincl, 63);
	const int maxl = wave_max_i32s(lane < n ? d_len : 0);
	// the difference array of one contig (4 B per base) sits at the tail of the block while the supports are counted
	const int scratch_b = (SB * C.sup_cap + 4 * C.pm_cap - 4 * (maxl + 2)) & ~15;     // (a byte offset from SUP)
	if (maxl > V3_MAXLEN || stotal > C.sup_cap || ptotal + 2 > C.pm_cap || SB * stotal > scratch_b) { V3_CNT(C, 8, 1); return IHP_E_CAPACITY; }
	if (lane < n) {
		S.dw[lane] = poff; S.so[lane] = d_nreads != 1 ? soff : -1; S.len[lane] = d_len; S.cap[lane] = align4(d_len); S.nreads[lane] = d_nreads;
		S.start[lane] = ((long long)d_shi << 32) | (unsigned)d_slo; S.sh[lane] = 0; S.listA[lane] = (short)lane;
		S.qt[lane] = 0; S.mt[lane] = 0;
	}
	C.clock = 0;
	C.bump_sup = stotal; C.bump_pm = ptotal;
	C.alive = n >= 64 ? ~0ull : (1ull << n) - 1ull;
	unsigned rc[NREC];                                               // nrr <= 64 NREC
	bool over = false;                                               // a support that does not fit its element
#pragma unroll
	for (int k = 0; k < NREC; ++k) rc[k] = 64 * k + lane < nrr ? REC[64 * k + lane] : 0xffffffffu;
	uint32_t *scratch = (uint32_t *)(C.SUP + scratch_b);
	for (int c = 0; c < n; ++c) {
		const int len = bcast(d_len, c), so = bcast(soff, c), nr = bcast(d_nreads, c), anchor = bcast(d_anchor, c);
		if (nr == 1) {                                           // a single read: support 1 everywhere, no bytes
			if (lane == 0) { S.smin[c] = 1; S.smax[c] = 1; S.lo3[c] = V3_NOZONE; S.hi3[c] = 0; S.loT[c] = 0; S.hiT[c] = (short)len; }
			continue;
		}
		for (int i = lane; i <= len; i += 64) scratch[i] = 0;
		LDS_ORDER();
		auto scatter = [&](unsigned rw) {
			if (rw != 0xffffffffu && (int)(rw & 63u) == c) {
				const int s = (int)((rw >> 6) & 0x7fffu) - 16384 + anchor, e = s + (int)(rw >> 21);
				atomicAdd(&scratch[s], 1u);
				atomicAdd(&scratch[e], 0xffffffffu);
			}
		};
#pragma unroll
		for (int k = 0; k < NREC; ++k) scatter(rc[k]);
		LDS_ORDER();
		unsigned carry = 0;
		SupStats st; st.init(v3_thr(nr));
		for (int i0 = 0; i0 < len; i0 += 64) {
			const int i = i0 + lane;
			unsigned v = i < len ? scratch[i] : 0u;
			v = wave_scan_add(v) + carry;
			if (i < len) { sup_st<ST>(C, so + i, v); over |= v > ST::SUP_MAX; }
			st.add(v, i, i < len);
			carry = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
		}
		st.store(S, c);
	}
	LDS_ORDER();
	if (ballot(over)) { V3_CNT(C, 8, 1); return IHP_E_CAPACITY; }  // 256 reads on one base: the byte-based passes (u32 supports) take the region
	// the packed bases as they are, one zero pad dword behind every contig (the scratch area overlapped PM: bases last)
	// Lane <-> dword of PM, 256 dwords a round with their loads in flight together (a loop over the contigs was one dependent
	// round trip to L2 per contig -- a dozen per region, each as long as a best_match phase).
	for (int g0 = 0; g0 < ptotal; g0 += 256) {
		int own[4] = {0, 0, 0, 0};
		for (int i = 1; i < n; ++i) {
			const int pi = __builtin_amdgcn_readlane(poff, i);
#pragma unroll
			for (int k = 0; k < 4; ++k) own[k] = g0 + 64 * k + lane >= pi ? i : own[k];
		}
		int off[4];                                                  // -1: a pad dword (or past the end)
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int g = g0 + 64 * k + lane;
			const int d = g - __builtin_amdgcn_ds_bpermute(own[k] << 2, poff);
			const int len = __builtin_amdgcn_ds_bpermute(own[k] << 2, d_len), so = __builtin_amdgcn_ds_bpermute(own[k] << 2, d_poff);
			off[k] = g < ptotal && 16 * d < len ? so + d : -1;
		}
		uint32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) v[k] = H[off[k] < 0 ? 0 : off[k]];   // every lane loads (the header when it has nothing to fetch): no branch between the loads
#pragma unroll
		for (int k = 0; k < 4; ++k) if (g0 + 64 * k + lane < ptotal) C.PM[g0 + 64 * k + lane] = off[k] < 0 ? 0u : v[k];
	}
	LDS_ORDER();
	return 0;
}

// ------------------------------------------------------------------------------------------------ trim
// trim(c, min_support) of contig.nim:49-68 on the u8 supports; only the slot's start / length move.
template <class ST>
__device__ inline void v3_trim(ST &S, const V3Ctx &C, int c, int ms)
{
	const int lane = lane_id();
	const int so = uni(S.so[c]), len = uni(S.len[c]);
	if (so < 0) return;                                              // support 1 everywhere: the caller never gets here with ms > 1 (ms <= nreads = 1)
	const typename ST::sup_t *sup = (const typename ST::sup_t *)C.SUP + so;
	int a = len - 1 > 0 ? len - 1 : 0;
	for (int b = 0; b < len - 1; b += 64) {
		const int i = b + lane;
		const unsigned long long m = ballot(i < len - 1 && (int)sup[i] >= ms);
		if (m) { a = b + ctz64(m); break; }
	}
	if (a >= len - 1) {                                              // :56-60
		if (lane == 0) { S.start[c] += a; S.len[c] = 0; S.nreads[c] = 0; }
		LDS_ORDER();
		return;
	}
	int bb = a;                                                      // :62-64
	for (int top = len - 1; top > a; top -= 64) {
		const int i = top - lane;
		const unsigned long long m = ballot(i > a && (int)sup[i] >= ms);
		if (m) { bb = top - ctz64(m); break; }
	}
	if (lane == 0) {
		const int b = S.sh[c] + a;
		S.start[c] += a; S.so[c] = so + a; S.cap[c] -= a; S.len[c] = bb - a + 1;
		S.dw[c] += b >> 4; S.sh[c] = (unsigned char)(b & 15);
	}
	LDS_ORDER();
}

// ------------------------------------------------------------------------------------------------ best_match
// The candidates of a pass in registers: lane j <-> contig in[j] (see CombDir in asm2_dev.h; same idea on V3State).
struct Dir3 {
	int ts, len, pb, nr, nit, mt; unsigned head, smin, smax, excl; // pb = base address of the contig's first base in PM; mt = V3State::mt
	int own0, own1, ib0, ib1;                                       // lane g <-> item g / 64 + g: owner lane, dword index of the item
	int Q, n;
	unsigned long long inout;
	bool valid;
};

template <class ST>
__device__ inline void dir3_build(const ST &S, const V3Ctx &C, const short *in, int n, int min_overlap, Dir3 &D)
{
	const int lane = lane_id();
	n = uni(n);
	D.n = n; D.ts = 0; D.len = 0; D.pb = 0; D.nr = 0; D.nit = 0; D.mt = 0; D.head = 0; D.smin = 0; D.smax = 0; D.excl = 0;
	D.own0 = D.own1 = D.ib0 = D.ib1 = 0; D.Q = 0;
	D.valid = n <= 64 && n > 0 && min_overlap >= 17;
	if (!D.valid) return;
	if (lane < n) {
		// (a contig shorter than 16 bases has a head that runs into its padding: harmless, an overlap with it is shorter than
		// min_overlap - 1 and is never a candidate)
		D.ts = in[lane]; D.len = S.len[D.ts]; D.pb = 16 * S.dw[D.ts] + S.sh[D.ts];
		D.head = pk16(C.PM, D.pb);
		D.smin = S.smin[D.ts]; D.smax = S.smax[D.ts]; D.nr = S.nreads[D.ts]; D.mt = S.mt[D.ts];
	}
	// items of the target-offset scan: the dwords of a contig that hold an offset 0 .. len - min_overlap
	D.nit = lane < n && D.len >= min_overlap ? (((D.pb & 15) + D.len - min_overlap) >> 4) + 1 : 0;
	const unsigned incl = wave_scan_add((unsigned)D.nit);
	D.excl = incl - (unsigned)D.nit;
	D.Q = __builtin_amdgcn_readlane((int)incl, 63);
	int own0 = 0, own1 = 0;                                          // items 0 .. 127 are mapped here; any beyond them when they are looked at
	for (int i = 0; i < n; ++i) {
		const int ex = __builtin_amdgcn_readlane((int)D.excl, i), ni = __builtin_amdgcn_readlane(D.nit, i);
		if (ni) { own0 = lane >= ex ? i : own0; own1 = 64 + lane >= ex ? i : own1; }
	}
	D.own0 = own0; D.own1 = own1;
	D.ib0 = (__builtin_amdgcn_ds_bpermute(own0 << 2, D.pb) >> 4) + (lane - __builtin_amdgcn_ds_bpermute(own0 << 2, (int)D.excl));
	D.ib1 = (__builtin_amdgcn_ds_bpermute(own1 << 2, D.pb) >> 4) + (64 + lane - __builtin_amdgcn_ds_bpermute(own1 << 2, (int)D.excl));
}

// Candidates of one best_match call: lane i <-> candidate i (key, base addresses of the two sides, overlap length).
struct Cand3 { unsigned key; int xb, yb, cn; int n; };
__device__ __forceinline__ unsigned cand_key(int cn, int pos, int ph, int o)
{   // smaller is better: more matches, then the earlier contig, then the target-offset phase, then the smaller offset
	return ((unsigned)(V3_MAXLEN - cn) << 18) | ((unsigned)pos << 12) | ((unsigned)ph << 11) | (unsigned)o;
}

// Verify all candidates together and fold the best passing one into `bestkey` (0xffffffff: none yet).
__device__ inline void cand_flush(const V3Ctx &C, Cand3 &K, unsigned &bestkey)
{
	const int lane = lane_id();
	const int nc = K.n;
	if (nc == 0) return;
	K.n = 0;
	// a candidate that cannot beat the best so far needs no look (keys are unique per (contig, phase, offset))
	const bool live = lane < nc && K.key < bestkey;
	const int nch = live ? (K.cn + 15) >> 4 : 0;
	const unsigned incl = wave_scan_add((unsigned)nch), excl = incl - (unsigned)nch;
	const int T = __builtin_amdgcn_readlane((int)incl, 63);
	V3_CNT(C, 1, nc); V3_CNT(C, 2, (T + 63) / 64);
	bool fail = false;
	for (int g0 = 0; g0 < T; g0 += 64) {
		const int g = g0 + lane;
		int own = 0;
		for (int i = 0; i < nc; ++i) own = ((unsigned)g >= (unsigned)__builtin_amdgcn_readlane((int)excl, i) && __builtin_amdgcn_readlane(nch, i)) ? i : own;
		const int ox = __builtin_amdgcn_ds_bpermute(own << 2, K.xb), oy = __builtin_amdgcn_ds_bpermute(own << 2, K.yb);
		const int ocn = __builtin_amdgcn_ds_bpermute(own << 2, K.cn), oex = __builtin_amdgcn_ds_bpermute(own << 2, (int)excl);
		bool bad = false;
		if (g < T) {
			const int k = g - oex, rem = ocn - 16 * k;
			unsigned x = pk16(C.PM, ox + 16 * k) ^ pk16(C.PM, oy + 16 * k);
			if (rem < 16) x &= (1u << (2 * rem)) - 1u;
			bad = x != 0;
		}
		const unsigned long long fm = ballot(bad);
		// candidate lane: did any of my chunks in this pass fail?
		const int s0 = (int)excl - g0, e0 = s0 + nch;
		const int lo = s0 < 0 ? 0 : s0, hi = e0 > 64 ? 64 : e0;
		if (live && hi > lo) fail |= (fm & lane_range64(lo, hi - 1)) != 0;
	}
	const unsigned k = live && !fail ? K.key : 0xffffffffu;
	const unsigned kmin = wave_min_u32(k);
	bestkey = kmin < bestkey ? kmin : bestkey;
}

__device__ __forceinline__ void cand_push(const V3Ctx &C, Cand3 &K, unsigned &bestkey, unsigned key, int xb, int yb, int cn)
{
	const bool sel = lane_id() == K.n;
	K.key = sel ? key : K.key; K.xb = sel ? xb : K.xb; K.yb = sel ? yb : K.yb; K.cn = sel ? cn : K.cn;
	if (++K.n == 64) cand_flush(C, K, bestkey);
}

struct Best3 { int found, ma, pos, slot, off, ord; };   // ord: place of the offset in the scan of its pair (ties between waves)

// slide_align (contig.nim:70-141) of contig qs on contig ts when the vote rule may fire for the pair, max_mismatch 0:
// an offset stands iff every differing base is an allowed mismatch; its matches are the equal bases.  Lanes are offsets
// for the filter: 16 bases where neither contig can be overruled (the [loT, hiT) runs of both) must be equal; an offset
// without such a window must have allowed mismatches among its first 16 bases.  Every survivor is then checked over its
// whole overlap by the wave (16 bases per lane; supports are only read where the bases differ).
// `best` is shared by the targets of one best_match call in list order: strictly more matches win (contig.nim:107, :239).
// (part, nparts, ctr): the 64-offset chunks of all vote pairs of a best_match call are dealt round robin to the waves that
// share the call (v3_best_match); ctr counts them.
template <class ST>
__device__ inline void v3_slide_votes(const ST &S, const V3Ctx &C, int qs, int ts, int pos, int min_overlap, Best3 &best,
                                      int part, int nparts, int &ctr)
{
	const int lane = lane_id();
	const int qlen = uni(S.len[qs]), tlen = uni(S.len[ts]);
	const int qpb = uni(16 * S.dw[qs] + S.sh[qs]), tpb = uni(16 * S.dw[ts] + S.sh[ts]);
	const int qso = uni(S.so[qs]), tso = uni(S.so[ts]);
	const int qreads = uni(S.nreads[qs]), treads = uni(S.nreads[ts]);
	const int qloT = uni((int)S.loT[qs]), qhiT = uni((int)S.hiT[qs]), tloT = uni((int)S.loT[ts]), thiT = uni((int)S.hiT[ts]);
	const int omax = tlen - min_overlap;                             // :79
	int omin_abs = qlen - min_overlap;                               // :78, :114
	if (omin_abs < 0) omin_abs = -omin_abs;
	const int n1 = omax >= 0 ? omax + 1 : 0;
	const int total = n1 + omin_abs;
	V3_CNT(C, 14, (total + 63) / 64);
	for (int base = 0; base < total; base += 64) {
		if (nparts > 1) { const bool mine = ctr == part; ctr = ctr + 1 == nparts ? 0 : ctr + 1; if (!mine) continue; }
		const int idx = base + lane;
		int qo0 = 0, to0 = 0;
		if (idx < n1) to0 = idx; else qo0 = idx - n1 + 1;
		int n = qlen - qo0 < tlen - to0 ? qlen - qo0 : tlen - to0;
		if (n < 0) n = 0;
		const int need = best.found && best.ma + 1 > min_overlap - 1 ? best.ma + 1 : min_overlap - 1;   // matches <= n
		bool surv = idx < total && n >= need;
		bool nozone = false;
		unsigned m = 0;
		if (surv) {
			int klo = qloT - qo0 > tloT - to0 ? qloT - qo0 : tloT - to0;
			if (klo < 0) klo = 0;
			int khi = qhiT - qo0 < thiT - to0 ? qhiT - qo0 : thiT - to0;
			if (khi > n) khi = n;
			if (khi - klo >= 16) {
				const int k = klo + ((khi - klo - 16) >> 1);             // the middle of the run both are strong on
				surv = pk16(C.PM, qpb + qo0 + k) == pk16(C.PM, tpb + to0 + k);
			} else {
				m = diff16(pk16(C.PM, qpb + qo0), pk16(C.PM, tpb + to0));
				if (n < 16) m &= (1u << (2 * n)) - 1u;
				nozone = true;
			}
		}
		// an offset without a window that both contigs are strong on: its differing bases at four places of the overlap
		// must be allowed ones (each probe is exact; most such offsets meet a base neither side can be overruled on)
		unsigned long long weak = ballot(surv && nozone);
		for (int pr = 0; pr < 4 && weak; ++pr) {
			if (surv && nozone) {
				const int k = pr == 0 ? 0 : n > 16 ? ((n - 16) * pr) / 3 : 0;
				unsigned mm = pr == 0 ? m : diff16(pk16(C.PM, qpb + qo0 + k), pk16(C.PM, tpb + to0 + k));
				if (pr && n < 16) mm &= (1u << (2 * n)) - 1u;
				for (int t = 0; t < 8 && mm && surv; ++t) {
					const int j = __builtin_ctz(mm) >> 1;
					mm &= mm - 1;
					surv = allowed3(sup_ld<ST>(C, qso + qo0 + k + j), sup_ld<ST>(C, tso + to0 + k + j), qreads, treads);
				}
			}
			weak = ballot(surv && nozone);
		}
		unsigned long long mask = ballot(surv);
		V3_CNT(C, 15, popc64(mask));
		while (mask) {
			const int sl = ctz64(mask);
			mask &= mask - 1;
			const int cq = bcast(qo0, sl), ct = bcast(to0, sl), cn = bcast(n, sl);
			if (best.found && cn <= best.ma) continue;               // cannot have more matches than the best so far
			int mmiss = 0;
			bool bad = false;
			for (int k0 = 0; k0 < cn && !bad; k0 += 1024) {
				const int k = k0 + 16 * lane;
				unsigned mm = 0;
				if (k < cn) {
					mm = diff16(pk16(C.PM, qpb + cq + k), pk16(C.PM, tpb + ct + k));
					const int rem = cn - k;
					if (rem < 16) mm &= (1u << (2 * rem)) - 1u;
				}
				mmiss += __popc(mm);
				bool lbad = false;
				if (ballot(mm != 0)) {
					while (mm && !lbad) {
						const int j = __builtin_ctz(mm) >> 1;
						mm &= mm - 1;
						lbad = !allowed3(sup_ld<ST>(C, qso + cq + k + j), sup_ld<ST>(C, tso + ct + k + j), qreads, treads);
					}
				}
				bad = ballot(lbad) != 0;
			}
			if (bad) continue;
			const int ma = cn - wave_sum_i(mmiss);
			if (ma >= min_overlap - 1 && (!best.found || ma > best.ma)) {
				best.found = 1; best.ma = ma; best.pos = pos; best.slot = ts; best.off = cq ? -cq : ct; best.ord = base + sl;
			}
		}
	}
}

// since > 0 (pass 2, a query that has not changed since it was the query of pass 1 at step `since`): only the contigs that
// have changed since then are looked at -- against the others slide_align found nothing then and would find nothing now.
// One wave's share (part of nparts) of a best_match call: the vote pairs' chunks, the target-offset turns and the query-offset
// chunks are dealt round robin; `bestkey` = its best exact candidate (0xffffffff: none), G = its best vote-pair offset.
// Reads LDS only.
template <class ST>
__device__ inline void v3_bm_part(const ST &S, const V3Ctx &C, const Dir3 &D, int qi, int min_overlap, int since,
                                  int part, int nparts, unsigned &bestkey_out, Best3 &G)
{
	const int lane = lane_id();
	qi = uni(qi);
	bestkey_out = 0xffffffffu;
	const int qs = __builtin_amdgcn_readlane(D.ts, qi), qlen = __builtin_amdgcn_readlane(D.len, qi);
	const int qpb = __builtin_amdgcn_readlane(D.pb, qi);
	const int omin = qlen - min_overlap;                             // contig.nim:78 (>= 0: the caller checked)
	const unsigned qh = (unsigned)__builtin_amdgcn_readlane((int)D.head, qi);
	const unsigned qmin = (unsigned)__builtin_amdgcn_readlane((int)D.smin, qi), qmax = (unsigned)__builtin_amdgcn_readlane((int)D.smax, qi);
	const int qreads = __builtin_amdgcn_readlane(D.nr, qi);
	const uint32_t *PM = C.PM;
	G = Best3{0, 0, -1, -1, 0, 0};                                   // best of the pairs that need the vote scan
	const bool in = lane_of(D.inout) && D.mt >= since;
	const unsigned long long inm = ballot(in);
	if (!inm) return;
	const bool votes = in && ((qmin < 3u && D.smax > 3u * qmin && qreads > 3 * (int)qmin) ||
	                          (D.smin < 3u && qmax > 3u * D.smin && D.nr > 3 * (int)D.smin));
	const unsigned long long vm = ballot(votes);
	{
		V3_T0(C);
		unsigned long long gm = vm;
		int ctr = 0;
		while (gm) {
			const int i = ctz64(gm);
			gm &= gm - 1;
			v3_slide_votes(S, C, qs, __builtin_amdgcn_readlane(D.ts, i), i, min_overlap, G, part, nparts, ctr);
			V3_CNT(C, 3, 1);
		}
		if (vm) V3_T1(C, 4);
	}
	V3_T0(C);
	long long tl_ = t0_;
#define V3_LAP(k) do { if (C.prof) { const long long t_ = (long long)clock64(); if (lane == 0) C.prof[k] += t_ - tl_; tl_ = t_; } } while (0)
	const unsigned long long usem = inm & ~vm;
	Cand3 K; K.key = 0; K.xb = K.yb = K.cn = 0; K.n = 0;
	unsigned bestkey = 0xffffffffu;
	V3_CNT(C, 0, 1); V3_CNT(C, 5, (__builtin_amdgcn_readlane((int)D.excl, qi) + 63) / 64); V3_CNT(C, 6, popc64(usem) * (omin / 64 + 1));
	// ---- offsets 0 .. len(t) - min_overlap on the contigs (:79-111): items = dwords that hold such an offset
	// (the contigs of `out` all come before the query in the list: only the items below the query's own are looked at)
	const int qlim = (int)__builtin_amdgcn_readlane((int)D.excl, qi);
	// lane <-> item 64 w + lane: its contig and the dword it stands for
	auto chunk_map = [&](int w, int &own, int &ib) {
		if (w == 0) { own = D.own0; ib = D.ib0; return; }
		if (w == 1) { own = D.own1; ib = D.ib1; return; }
		// long contigs, more than 128 items.  Items are in contig order: only the contigs whose items reach into
		// [64 w, 64 w + 63] can own one of them (two or three of them, whatever the number of contigs)
		const int g = 64 * w + lane;
		const unsigned long long below = (1ull << qi) - 1ull;
		const unsigned long long has = ballot(D.nit > 0 && (int)D.excl + D.nit > 64 * w) & below;
		const unsigned long long beg = ballot(D.nit > 0 && (int)D.excl <= 64 * w + 63) & below;
		const int c_lo = has ? ctz64(has) : 0, c_hi = beg ? 63 - clz64(beg) : -1;
		own = c_lo; ib = 0;
		for (int i = c_lo; i <= c_hi; ++i) {
			const int ex = __builtin_amdgcn_readlane((int)D.excl, i), ni = __builtin_amdgcn_readlane(D.nit, i);
			const int d0 = (__builtin_amdgcn_readlane(D.pb, i) >> 4) - ex;
			if (ni) { const bool t = g >= ex; own = t ? i : own; ib = t ? d0 + g : ib; }
		}
	};
	auto chunk_hits = [&](int own, int ib, unsigned w0, unsigned w1, unsigned long long hm) {
		V3_CNT(C, 11, popc64(hm));
		while (hm) {
			const int e = ctz64(hm);
			hm &= hm - 1;
			unsigned bits = window_bits((unsigned)__builtin_amdgcn_readlane((int)w0, e), (unsigned)__builtin_amdgcn_readlane((int)w1, e), qh);
			V3_CNT(C, 12, __popc(bits));
			const int i = __builtin_amdgcn_readlane(own, e);
			const int tlen = __builtin_amdgcn_readlane(D.len, i), tpb = __builtin_amdgcn_readlane(D.pb, i);
			const int ibe = __builtin_amdgcn_readlane(ib, e);
			while (bits) {
				const int o = 16 * ibe + __builtin_ctz(bits) - tpb;    // base address of the window minus the contig's first base
				bits &= bits - 1;
				if (o < 0 || o > tlen - min_overlap) continue;
				const int cn = qlen < tlen - o ? qlen : tlen - o;
				cand_push(C, K, bestkey, cand_key(cn, i, 0, o), qpb, tpb + o, cn);
			}
		}
	};
	// two chunks a turn: their four loads are in flight together
	// (a wave that shares the call takes every nparts-th chunk)
	const int nw = (qlim + 63) >> 6;
	for (int w = part; w < nw; w += 2 * nparts) {
		const int w2 = w + nparts;
		int ownA, ibA, ownB = 0, ibB = 0;
		chunk_map(w, ownA, ibA);
		const bool two = w2 < nw;
		if (two) chunk_map(w2, ownB, ibB);
		const bool actA = 64 * w + lane < qlim && ((usem >> ownA) & 1ull);
		const bool actB = two && 64 * w2 + lane < qlim && ((usem >> ownB) & 1ull);
		if (!ballot(actA || actB)) continue;
		unsigned a0 = 0, a1 = 0, b0 = 0, b1 = 0;
		if (actA) { a0 = PM[ibA]; a1 = PM[ibA + 1]; }
		if (actB) { b0 = PM[ibB]; b1 = PM[ibB + 1]; }
		const bool anyA = actA && window_any(a0, a1, qh), anyB = actB && window_any(b0, b1, qh);
		const unsigned long long hA = ballot(anyA), hB = ballot(anyB);
		if (hA) chunk_hits(ownA, ibA, a0, a1, hA);
		if (hB) chunk_hits(ownB, ibB, b0, b1, hB);
	}
	V3_LAP(9);
	// ---- offsets 1 .. omin on the query (:114-135): lane <-> offset, the contigs' first 16 bases come by
	for (int ob = 64 * part; ob <= omin; ob += 64 * nparts) {
		const int o_l = ob + lane;
		const bool valid = o_l >= 1 && o_l <= omin;
		const unsigned wq = valid ? pk16(PM, qpb + o_l) : 0u;
		const unsigned long long okm = ballot(valid);
		unsigned long long um = usem;
		while (um) {
			const int i = ctz64(um);
			um &= um - 1;
			unsigned long long mask = ballot(wq == (unsigned)__builtin_amdgcn_readlane((int)D.head, i)) & okm;
			if (!mask) continue;                                     // the usual case: the contig does not start inside the query
			const int tlen = __builtin_amdgcn_readlane(D.len, i), tpb = __builtin_amdgcn_readlane(D.pb, i);
			while (mask) {
				const int o = ob + ctz64(mask);
				mask &= mask - 1;
				const int cn = qlen - o < tlen ? qlen - o : tlen;
				if (cn < min_overlap - 1) continue;                  // best_ma starts at min_overlap - 1 (:81, :107)
				cand_push(C, K, bestkey, cand_key(cn, i, 1, o), qpb + o, tpb, cn);
			}
		}
	}
	V3_LAP(10);
	cand_flush(C, K, bestkey);
	V3_LAP(11);
#undef V3_LAP
	V3_T1(C, 5);
	bestkey_out = bestkey;
}

// of two vote-pair results the one slide_align / best_match would have kept scanning everything in order: more matches, then
// the earlier contig, then the earlier offset of the pair's scan (contig.nim:107 replaces on strictly more matches only)
__device__ __forceinline__ bool best3_before(const Best3 &a, const Best3 &b)
{
	if (!a.found || !b.found) return a.found != 0;
	if (a.ma != b.ma) return a.ma > b.ma;
	if (a.pos != b.pos) return a.pos < b.pos;
	return a.ord < b.ord;
}

__device__ inline Best3 v3_bm_finish(const Dir3 &D, unsigned bestkey, const Best3 &G)
{
	Best3 B = {0, 0, -1, -1, 0, 0};
	if (bestkey != 0xffffffffu) {
		const int o = (int)(bestkey & 2047u), ph = (int)((bestkey >> 11) & 1u), pos = (int)((bestkey >> 12) & 63u);
		B.found = 1; B.ma = V3_MAXLEN - (int)(bestkey >> 18); B.pos = pos; B.slot = __builtin_amdgcn_readlane(D.ts, pos); B.off = ph ? -o : o;
	}
	// more matches, then (no mismatches either way) the earlier contig (contig.nim:32-36, :107, :239)
	if (G.found && (!B.found || G.ma > B.ma || (G.ma == B.ma && G.pos < B.pos))) return G;
	return B;
}

// ---- several waves on one best_match call -----------------------------------------------------------------------------
// A region's combine is one serial chain in ONE wave (wave 0 of the workgroup: everything that writes the region's state),
// and the LDS a region needs fixes 8-16 regions per CU, so each SIMD hosts two to four such waves and issues a fraction of
// what it could.  The other waves of the workgroup wait at a barrier; for a best_match call with enough to look at, wave 0
// leaves the call's parameters in LDS, everybody passes the barrier, takes its share of the call (v3_bm_part reads LDS only:
// packed bases, supports, the contig table), leaves its result in a mailbox and meets at a second barrier; wave 0 folds the
// mailboxes.  Between two calls only wave 0 runs, so nothing it writes is read half-done.
constexpr int V3_MAXW = 4;
struct V3Par {
	int cmd;                               // 1: a call; 2: the launch is over
	int list_b, n, min_overlap, qi, since, dver;
	unsigned long long inout;
	unsigned key[V3_MAXW];
	Best3 g[V3_MAXW];
};
struct V3Team { V3Par *par; int nparts; int list_b, n; int dver; };   // wave 0's side: dver counts the dir3_build calls

template <bool TEAM, class ST>
__device__ inline Best3 v3_best_match(const ST &S, const V3Ctx &C, const Dir3 &D, int qi, int min_overlap, int since, const V3Team &T)
{
	const int lane = lane_id();
	unsigned key; Best3 G;
	if (!TEAM) { v3_bm_part(S, C, D, qi, min_overlap, since, 0, 1, key, G); return v3_bm_finish(D, key, G); }
	// worth waking the others for?  target-offset turns + query-offset looks + vote pairs of the call
	bool team = false;
	if (T.nparts > 1) {
		const int qlim = (int)__builtin_amdgcn_readlane((int)D.excl, qi), qlen = __builtin_amdgcn_readlane(D.len, qi);
		const int work = (qlim >> 7) + ((qlen - min_overlap) >> 6) * (popc64(D.inout) >> 3);
		team = work >= 3;
	}
	const int np = team ? T.nparts : 1;
	V3Par *P = T.par;
	if (team) {
		if (lane == 0) {
			P->cmd = 1; P->list_b = T.list_b; P->n = T.n; P->min_overlap = min_overlap; P->qi = qi; P->since = since; P->dver = T.dver;
			P->inout = D.inout;
		}
		__syncthreads();
	}
	v3_bm_part(S, C, D, qi, min_overlap, since, 0, np, key, G);
	if (!team) return v3_bm_finish(D, key, G);
	__syncthreads();
	for (int w = 1; w < T.nparts; ++w) {
		const unsigned kw = (unsigned)uni((int)P->key[w]);
		key = kw < key ? kw : key;
		Best3 gw;
		gw.found = uni(P->g[w].found); gw.ma = uni(P->g[w].ma); gw.pos = uni(P->g[w].pos); gw.slot = uni(P->g[w].slot);
		gw.off = uni(P->g[w].off); gw.ord = uni(P->g[w].ord);
		if (best3_before(gw, G)) G = gw;
	}
	return v3_bm_finish(D, key, G);
}

// waves 1 .. of the workgroup: see V3Par
template <class ST>
__device__ inline void v3_helper_loop(const ST &S, const V3Ctx &C, V3Par *P, int part, int nparts)
{
	const int lane = lane_id();
	Dir3 D;
	D.valid = false;
	int dver = -1;
	for (;;) {
		__syncthreads();
		if (uni(P->cmd) != 1) break;
		const int mo = uni(P->min_overlap);
		if (uni(P->dver) != dver) {
			dir3_build(S, C, uni(P->list_b) ? S.listB : S.listA, uni(P->n), mo, D);
			dver = uni(P->dver);
		}
		D.inout = (unsigned long long)uni((long long)P->inout);
		unsigned key; Best3 G;
		v3_bm_part(S, C, D, uni(P->qi), mo, uni(P->since), part, nparts, key, G);
		if (lane == 0) { P->key[part] = key; P->g[part] = G; }
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ corrections + insert
template <class ST>
__device__ inline void v3_compact(ST &S, V3Ctx &C);

// The allowed mismatches of (q, t, offset) in scan order to the top of the packed area, counting down (contig.nim:99, :128).  Returns the count or -1.
template <class ST>
__device__ inline int v3_corrections(ST &S, V3Ctx &C, int qs, int ts, int off)
{
	const int lane = lane_id();
	if (C.pm_cap - C.bump_pm < 160) v3_compact(S, C);                // the corrections go to the top of the packed area: make room there first
	const int qlen = uni(S.len[qs]), tlen = uni(S.len[ts]);
	const int qpb = uni(16 * S.dw[qs] + S.sh[qs]), tpb = uni(16 * S.dw[ts] + S.sh[ts]);
	const int qso = uni(S.so[qs]), tso = uni(S.so[ts]);
	const int qo0 = off < 0 ? -off : 0, to0 = off < 0 ? 0 : off;
	const int n = qlen - qo0 < tlen - to0 ? qlen - qo0 : tlen - to0;
	int cnt = 0;
	constexpr int CW = ST::CORR_DW;
	int lim = (C.pm_cap - C.bump_pm - 4) / CW;                       // corrections that fit the free dwords above everything that is allocated
	lim = lim < V3_CORR ? lim : V3_CORR;
	for (int k0 = 0; k0 < n; k0 += 1024) {
		const int k = k0 + 16 * lane;
		unsigned m = 0;
		if (k < n) {
			m = diff16(pk16(C.PM, qpb + qo0 + k), pk16(C.PM, tpb + to0 + k));
			const int rem = n - k;
			if (rem < 16) m &= (1u << (2 * rem)) - 1u;
		}
		if (!ballot(m != 0)) continue;
		// with max_mismatch 0 every difference of the accepted offset is an allowed one
		const int mine = __popc(m);
		const unsigned incl = wave_scan_add((unsigned)mine);
		int w = cnt + (int)incl - mine;
		while (m) {
			const int j = __builtin_ctz(m) >> 1;
			m &= m - 1;
			const unsigned a = sup_ld<ST>(C, qso + qo0 + k + j), b = sup_ld<ST>(C, tso + to0 + k + j);
			// qoff | toff << 11 | qbest << 22 (| t's final support at the site << 23 once v3_insert has applied it)
			if (w < lim) C.PM[C.pm_cap - 1 - CW * w] = (unsigned)(qo0 + k + j) | ((unsigned)(to0 + k + j) << 11) | ((a > b ? 1u : 0u) << 22);
			++w;
		}
		cnt += __builtin_amdgcn_readlane((int)incl, 63);
	}
	LDS_ORDER();
	return cnt <= lim ? cnt : -1;
}

// Close the holes of both areas: the live contigs, in slot order, move down to the start (ascending copies, 64 elements
// at a time through registers, so a slot may overlap its own old place).
template <class ST>
__device__ inline void v3_compact(ST &S, V3Ctx &C)
{
	const int lane = lane_id();
	V3_CNT(C, 13, 1);
	const bool live = lane_of(C.alive);
	int key = live ? S.dw[lane] : 0x7fffffff;                        // both areas were filled in the same order
	int nsup = 0, npm = 0;
	for (;;) {
		const int k = wave_min_i32(key);
		if (k == 0x7fffffff) break;
		const int c = ctz64(ballot(key == k));
		key = lane == c ? 0x7fffffff : key;
		const int so = uni(S.so[c]), dw = uni(S.dw[c]), sh = uni((int)S.sh[c]), len = uni(S.len[c]);
		// dwords: the bases from `sh` on, room for the rest of the capacity the contig keeps (an insert in place may grow it to
		// cap bases without asking -- with sh > 0 that can be a dword more than the bases it has now), and the pad
		// -- never more than the capacity it has (a trim at the front leaves cap = old cap - a, which need not be a multiple of
		// four: rounding len up past it would ask for a dword the slot never had, and the contigs behind it would have to move
		// UP through each other).  So every contig moves down or stays.
		const int cap0 = uni(S.cap[c]);
		const int capn = align4(len) < cap0 ? align4(len) : cap0;
		const int nsrc = (sh + len + 15) >> 4, nd = ((sh + capn + 15) >> 4) + 1;
		if (so >= 0 && so != nsup) {
			for (int i0 = 0; i0 < len; i0 += 64) {
				const int i = i0 + lane;
				const unsigned v = i < len ? sup_ld<ST>(C, so + i) : 0u;
				LDS_ORDER();
				if (i < len) sup_st<ST>(C, nsup + i, v);
				LDS_ORDER();
			}
		}
		if (dw != npm) {
			for (int i0 = 0; i0 < nd; i0 += 64) {
				const int i = i0 + lane;
				const unsigned v = i < nsrc ? C.PM[dw + i] : 0u;
				LDS_ORDER();
				if (i < nd) C.PM[npm + i] = v;
				LDS_ORDER();
			}
		}
		if (lane == 0) { S.so[c] = so >= 0 ? nsup : -1; S.dw[c] = npm; S.cap[c] = capn; }
		if (so >= 0) nsup += align4(len) + SLOT_PAD;
		npm += nd;
	}
	C.bump_sup = nsup; C.bump_pm = npm;
	LDS_ORDER();
}

__device__ __forceinline__ bool v3_room(const V3Ctx &C, int ncap, int ncorr_dw)
{   // (the corrections of the merge sit at the top of the packed area until the insert is done: ncorr_dw dwords)
	return C.bump_sup + ncap + SLOT_PAD <= C.sup_cap && C.bump_pm + ((ncap + 15) >> 4) + 2 + ncorr_dw <= C.pm_cap;
}

__device__ __forceinline__ unsigned pk_base(const uint32_t *PM, int b) { return (PM[b >> 4] >> (2 * (b & 15))) & 3u; }

// insert(t, q, m) of contig.nim:156-222 with the ncorr corrections v3_corrections left at the top of the packed area.  q is not kept up to date (combine
// drops it right after: its corrected bases and supports are never looked at again).  Leaves t's support extrema / zone.
template <class ST>
__device__ inline int v3_insert(ST &S, V3Ctx &C, int ts, int qs, int off, int ncorr)
{
	const int lane = lane_id();
	const int qlen = uni(S.len[qs]), tlen = uni(S.len[ts]);
	const int aoff = off < 0 ? -off : off;
	int newlen;
	if (off < 0) { newlen = aoff + tlen; if (qlen > newlen) newlen = qlen; }
	else { newlen = tlen; if (off + qlen > newlen) newlen = off + qlen; }
	if (newlen > V3_MAXLEN) return IHP_E_CAPACITY;
	// (a target of one read has no support bytes: the merged contig gets a slot of its own whatever the offset)
	const bool reloc = off < 0 || newlen > uni(S.cap[ts]) || uni(S.so[ts]) < 0;
	int ncap = align4(newlen + headroom(newlen));
	constexpr int CW = ST::CORR_DW;
	if (reloc && !v3_room(C, ncap, CW * ncorr)) {
		v3_compact(S, C);                                            // (before any address of q or t is taken)
		if (!v3_room(C, ncap, CW * ncorr)) {
			ncap = align4(newlen);
			if (!v3_room(C, ncap, CW * ncorr)) { V3_CNT(C, 9, 1); return IHP_E_CAPACITY; }
		}
	}
	const int qpb = uni(16 * S.dw[qs] + S.sh[qs]), qso = uni(S.so[qs]);
	int tpb = uni(16 * S.dw[ts] + S.sh[ts]), tso = uni(S.so[ts]);
	// ---- corrections (:161-173): the winner's base and support go to the loser; t's value at such a site is final
	for (int c = lane; c < ncorr; c += 64) {                       // (corrections come from vote scans: both contigs have support bytes)
		const unsigned cr = C.PM[C.pm_cap - 1 - CW * c];
		const int qoff = (int)(cr & 2047u), toff = (int)((cr >> 11) & 2047u);
		const bool qbest = (cr >> 22) & 1u;
		unsigned val = sup_ld<ST>(C, tso + toff);
		if (qbest) {
			const unsigned b = pk_base(C.PM, qpb + qoff);
			const int tb = tpb + toff;
			atomicAnd(&C.PM[tb >> 4], ~(3u << (2 * (tb & 15))));
			atomicOr(&C.PM[tb >> 4], b << (2 * (tb & 15)));
			val = sup_ld<ST>(C, qso + qoff);
			sup_st<ST>(C, tso + toff, val);
		}
		// the site keeps this support whatever is added below (:198, :217)
		if (CW == 1) C.PM[C.pm_cap - 1 - c] = (cr & 0x7fffffu) | (val << 23);
		else C.PM[C.pm_cap - 2 - CW * c] = val;
	}
	LDS_ORDER();
	int ndw = tpb >> 4, nsh = tpb & 15, nso = tso;
	if (reloc) {
		ndw = C.bump_pm; nsh = 0; nso = C.bump_sup;
		const int nw = (ncap + 15) >> 4;
		for (int i = lane; i <= nw; i += 64) C.PM[ndw + i] = 0;
		LDS_ORDER();
		if (off < 0) {                                               // :180-195
			copy_bits<true>(C.PM, ndw, 0, qpb >> 4, 2 * (qpb & 15), 2 * aoff);
			LDS_ORDER();
			copy_bits<true>(C.PM, ndw, 2 * aoff, tpb >> 4, 2 * (tpb & 15), 2 * tlen);
			LDS_ORDER();
			copy_bits<true>(C.PM, ndw, 2 * (aoff + tlen), qpb >> 4, 2 * ((qpb & 15) + aoff + tlen), 2 * (qlen - aoff - tlen));
		} else {
			copy_bits<true>(C.PM, ndw, 0, tpb >> 4, 2 * (tpb & 15), 2 * tlen);
			LDS_ORDER();
			copy_bits<true>(C.PM, ndw, 2 * tlen, qpb >> 4, 2 * ((qpb & 15) + tlen - off), 2 * (newlen - tlen));   // :220-221
		}
		C.bump_pm += nw + 1; C.bump_sup += ncap + SLOT_PAD;
	} else if (newlen > tlen) {
		// in place: the new bases behind t's last one; the slot's dwords past the old end may hold anything
		const int d0 = ndw, b0 = 2 * (nsh + tlen);
		copy_bits<true>(C.PM, d0, b0, qpb >> 4, 2 * ((qpb & 15) + tlen - off), 2 * (newlen - tlen));
		LDS_ORDER();
		// zero the tail of the last dword and the pad dword (windows read one dword past the end)
		const int endb = nsh + newlen;
		if (lane == 0) {
			if (endb & 15) C.PM[d0 + (endb >> 4)] &= (1u << (2 * (endb & 15))) - 1u;
			C.PM[d0 + ((endb + 15) >> 4)] = 0;
		}
	}
	LDS_ORDER();
	// ---- supports: new[i] = T(i) + Q(i) (:198-200, :216-219), then the corrected sites get their final value back
	const int tshift = off < 0 ? aoff : 0, qshift = off < 0 ? 0 : off;
	typedef typename ST::sup_t sup_t;
	const sup_t *ts_ = (const sup_t *)C.SUP + tso, *qs_ = (const sup_t *)C.SUP + qso;
	sup_t *ns_ = (sup_t *)C.SUP + nso;
	const int lo = reloc ? 0 : off, hi = reloc ? newlen : (off + qlen < newlen ? off + qlen : newlen);
	bool over = false;
	for (int i = lo + lane; i < hi; i += 64) {
		const int ti = i - tshift, qi = i - qshift;
		unsigned v = (ti >= 0 && ti < tlen) ? (tso >= 0 ? ts_[ti] : 1u) : 0u;
		if (qi >= 0 && qi < qlen) v += qso >= 0 ? qs_[qi] : 1u;
		over |= v > ST::SUP_MAX;
		ns_[i] = (sup_t)v;
	}
	// (only a region of 256 reads can get here: the sum of two contigs' supports on a base is at most the reads of the region.
	// A sum at a corrected site is replaced below and would not matter; the region is handed over all the same.)
	if (ballot(over)) { V3_CNT(C, 8, 1); return IHP_E_CAPACITY; }
	LDS_ORDER();
	for (int c = lane; c < ncorr; c += 64) {
		const unsigned cr = C.PM[C.pm_cap - 1 - CW * c];
		const unsigned fin = CW == 1 ? cr >> 23 : C.PM[C.pm_cap - 2 - CW * c];
		ns_[off < 0 ? (int)(cr & 2047u) : (int)((cr >> 11) & 2047u)] = (sup_t)fin;   // index in the merged contig (:170-173)
	}
	LDS_ORDER();
	const int nreads_new = uni(S.nreads[ts]) + uni(S.nreads[qs]);  // :203, :222
	SupStats st; st.init(v3_thr(nreads_new));
	for (int i0 = 0; i0 < newlen; i0 += 64) { const int i = i0 + lane; st.add(i < newlen ? ns_[i] : 0u, i, i < newlen); }
	if (lane == 0) {
		S.dw[ts] = ndw; S.sh[ts] = (unsigned char)nsh; S.so[ts] = nso; S.len[ts] = newlen;
		if (reloc) S.cap[ts] = ncap;
		S.nreads[ts] = nreads_new;
		if (off < 0) S.start[ts] = S.start[qs];                      // :204
	}
	st.store(S, ts);
	return 0;
}

// ------------------------------------------------------------------------------------------------ one pass of combine
// contig.nim:263-281: `in` -> `out`, returns the new count or < 0.
template <bool TEAM, class ST>
__device__ inline int v3_combine_pass(ST &S, V3Ctx &C, short *in, int n, short *out, int min_support, int min_overlap, V3Team &T)
{
	const int lane = lane_id();
	int nout = 0, usedi = 0;
	for (int i = 0; i < n; ++i) {                                    // :265-271
		const int c = uni((int)in[i]);
		if (min_support > 0) {
			V3_T0(C);
			const int nr = uni(S.nreads[c]);
			const int ms = nr < min_support ? nr : min_support;
			const int len0 = uni(S.len[c]), lo3 = uni((int)S.lo3[c]), hi3 = uni((int)S.hi3[c]);
			const long long start0 = uni(S.start[c]);
			if ((int)uni((int)S.smin[c]) >= ms && len0 >= 2) {
				// every support >= ms on two or more bases: trim (contig.nim:49-68) keeps the contig as it is
			} else if (ms == 3 && hi3 > lo3 && len0 >= 2) {
				// the supports >= 3 are one run [lo3, hi3): a = lo3 (first i < len - 1), b = hi3 - 1 (last i > a)
				if (lo3 >= len0 - 1) {                               // :56-60
					if (lane == 0) { S.start[c] += len0 - 1; S.len[c] = 0; S.nreads[c] = 0; }
				} else {
					const int a0 = lo3, b0 = hi3 - 1 > lo3 ? hi3 - 1 : lo3;
					if (lane == 0) {
						const int b = S.sh[c] + a0;
						S.start[c] += a0; S.so[c] += a0; S.cap[c] -= a0; S.len[c] = b0 - a0 + 1;    // (a contig with a ">= 3" run has support bytes)
						S.dw[c] += b >> 4; S.sh[c] = (unsigned char)(b & 15);
						S.smin[c] = 3; S.lo3[c] = 0; S.hi3[c] = (short)(b0 - a0 + 1);   // what is left has every support >= 3 (a bound will do for smin)
						S.loT[c] = 0; S.hiT[c] = (short)(b0 - a0 + 1);
					}
				}
				LDS_ORDER();
			} else {
				v3_trim(S, C, c, ms);
				v3_stats(S, C, c);
				V3_CNT(C, 7, 1);
			}
			if (uni(S.len[c]) != len0 || uni(S.start[c]) != start0) { if (lane == 0) S.mt[c] = (short)(C.clock + 1); LDS_ORDER(); }
			V3_T1(C, 7);
		}
		if (uni(S.nreads[c]) > 0 && nout == 0) {
			if (lane == 0) out[0] = (short)c;
			nout = 1; usedi = i;
		}
	}
	LDS_ORDER();
	if (nout == 0) return 0;                                         // :272
	Dir3 D;
	dir3_build(S, C, in, n, min_overlap, D);
	if (!D.valid) return IHP_E_CAPACITY;
	T.list_b = in == S.listB; T.n = n; T.dver++;
	D.inout = 1ull << usedi;
	for (int i = 0; i < n; ++i) {                                    // :274-281
		if (i == usedi) continue;
		const int c = uni((int)in[i]);
		Best3 b = {0, 0, -1, -1, 0, 0};
		C.clock++;
		int since = 0;
		if (min_support == 0) { if (lane == 0) S.qt[c] = (short)C.clock; }
		else { const int qt = uni((int)S.qt[c]); if (qt > 0 && uni((int)S.mt[c]) < qt) since = qt; }
		{
			// a query shorter than min_overlap - 1 cannot reach min_overlap - 1 matches at any offset (contig.nim:107); one of
			// exactly that length can, at a target offset only (omin = -1: the abs() of contig.nim:114 gives one query offset,
			// whose overlap is one base short)
			const int ql = __builtin_amdgcn_readlane(D.len, i);
			if (ql >= min_overlap - 1) b = v3_best_match<TEAM>(S, C, D, i, min_overlap, since, T);
		}
		if (b.found) {
			V3_T0(C);
			const int nc = v3_corrections(S, C, c, b.slot, b.off);
			if (nc < 0) { V3_CNT(C, 10, 1); return IHP_E_CAPACITY; }
			const int rc = v3_insert(S, C, b.slot, c, b.off, nc);
			if (rc) return rc;
			C.alive &= ~(1ull << c);                                 // q is gone (contig.nim:279)
			if (lane == 0) S.mt[b.slot] = (short)C.clock;
			LDS_ORDER();
			V3_CNT(C, 4, 1);
			const unsigned long long keep = D.inout;
			dir3_build(S, C, in, n, min_overlap, D);                 // the target changed: its lane and the item map again
			T.dver++;
			D.inout = keep;
			V3_T1(C, 6);
		} else if (uni(S.nreads[c]) > 0) {
			if (lane == 0) out[nout] = (short)c;
			nout++;
			D.inout |= 1ull << i;
		} else C.alive &= ~(1ull << c);                              // trimmed to nothing: dropped (:280)
		LDS_ORDER();
	}
	return nout;
}

}  // namespace ihp
