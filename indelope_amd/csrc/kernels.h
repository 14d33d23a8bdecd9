// kernels.h -- the __global__ entry points (gfx950).  Every kernel is a persistent
// grid of single-wavefront workgroups pulling work items off sharded counters (wq_next) or taking them round robin.
#pragma once
#include "contig_dev.h"
#include "ksw_dev.h"
#include "ksw_narrow.h"
#include "ksw_pair.h"
#include "ksw_duo.h"
#include "ksw_wide.h"
#include "tally_dev.h"
#include "roi_dev.h"

namespace ihp {

// ------------------------------------------------------------------ helpers of the ksw2 plan (k_ksw_plan_*, below)
constexpr int PLAN_KEYS = 4096;
// the contig length under which a job may be paired (ksw_pair.h), or -1; lds_budget / p_cap: of the pair launch
__device__ __forceinline__ int plan_key(const KswParams &P, int lds_budget, size_t p_cap, int qlen, int tlen, int flags)
{
	if (!(flags & ALN_Q_ACGT) || qlen >= PLAN_KEYS || !ksw_pair_job_ok(P, qlen, tlen)) return -1;
	if (2 * ksw_pair_lds_share(qlen, tlen) > (size_t)lds_budget || ksw_pair_p_bytes(qlen, P.w) > p_cap) return -1;
	return qlen;
}
// position of every lane with `take` in a list whose length is *cnt (one atomic per wavefront); -1 for the other lanes
__device__ __forceinline__ int plan_append(int *cnt, bool take)
{
	const unsigned long long m = ballot(take);
	if (!m) return -1;
	const int lane = lane_id();
	int base = 0;
	if (lane == ctz64(m)) base = atomicAdd(cnt, popc64(m));
	base = __builtin_amdgcn_readlane(base, ctz64(m));
	return take ? base + popc64(m & ((1ull << lane) - 1)) : -1;
}


// ------------------------------------------------------------------ assemble
struct AsmArgs {
	int n_regions;
	const long long *region_read_off, *read_off;
	const uint8_t *bases, *quals;
	const int *trim_lo, *trim_hi;                      // optional: trim() done by the stager (then quals is null)
	const long long *read_start, *read_stop;
	const uint8_t *mapq, *read_skip;
	const long long *ref_off, *ref_origin;
	double min_overlap_pct;
	int min_mapq_assemble, min_mapq_stop, trim_min_qual, combine_min_support, combine_min_overlap;
	int max_mismatch, max_pre_contigs, min_ctg_len, min_reads, K, ref_pad;
	// per-workgroup scratch (strides in elements)
	uint8_t *arena_seq; uint32_t *arena_sup; int arena_cap, stage_cap;
	Corr *corr; int corr_cap;
	// outputs
	int *status, *n_pre, *n_final;                     // [R]
	long long *ctg_start, *ctg_nreads, *ctg_seq_off;   // [slots]
	int *ctg_len, *aln_flags, *aln_ref_len;
	long long *aln_ref_start;
	uint8_t *out_seq; uint32_t *out_sup;
	AlnJob *jobs; int *n_jobs;
	int *work_counter;
	const int *in_list, *n_in;                         // regions to process (null: all n_regions)
	int *out_list, *n_out;                             // regions handed to the next, roomier pass (null: none)
	int lds_arena;                                     // bytes of the dynamic LDS arena (LDS passes)
	long long *prof;                                   // optional cycle counters (diagnostics)
	unsigned long long *t_start;                       // optional: wall clock at which the launch's first workgroup starts
	// packed read phase (asm2_dev.h): what k_prepack left -- 2-bit bases, the kept range of every read, "not ACGT" flags
	const uint32_t *v2_pk; const int *v2_trim_lo, *v2_trim_hi; const uint8_t *v2_read_bad;
	int v2_pdw;                                        // dwords of the per-wave packed area in LDS (k_asm_reads)
	int v2_pm_dw;                                      // dwords of the packed mirror behind the byte arena (k_asm_combine3)
	uint32_t *v2_hand; const long long *v2_hoff;       // hand-over records k_asm_reads -> k_asm_combine3: region r at v2_hand + v2_hoff[r]
	const int *lpt_cnt, *lpt_seg; int lpt_stride;      // k_asm_combine3: its regions by cost class (asm2_dev.h lpt_class; the counters and list
	int lpt_nclass;                                    // segments of this launch's tier -- or tiers: lpt_nclass of them, laid end to end), or null
};

}  // namespace ihp
#include "asm3_dev.h"
namespace ihp {

// Kernel execution time without a profiler: workgroup 0 is dispatched first, so its entry time is when the launch starts
// to execute (an event recorded before the launch fires when the stream is ready, which is earlier when another stream's
// kernels still hold the wave slots); a one-thread marker kernel behind the launch reads the clock when it has finished.
__device__ __forceinline__ void mark_start(unsigned long long *t)
{
	if (t && blockIdx.x == 0 && threadIdx.x == 0) *t = (unsigned long long)wall_clock64();
}
__global__ void k_mark(unsigned long long *t) { *t = (unsigned long long)wall_clock64(); }

// indelope.nim:23-38 on one read, lanes over bases.  Returns a; kept range [lo,hi).
__device__ inline int read_trim_dev(const uint8_t *q, int n, int min_quality, int &lo, int &hi)
{
	const int lane = lane_id();
	const int high = n - 1;
	const uint8_t mq = (uint8_t)min_quality;
	int a = high > 0 ? high : 0;
	for (int b = 0; b < high; b += 64) {                       // :25
		const int i = b + lane;
		const unsigned long long m = ballot(i < high && q[i] >= mq);
		if (m) { a = b + ctz64(m); break; }
	}
	if (a == high) { lo = 0; hi = 0; return a; }               // :28-30
	int bb = a;
	for (int top = high; top > a; top -= 64) {                 // :33
		const int i = top - lane;
		const unsigned long long m = ballot(i > a && q[i] >= mq);
		if (m) { bb = top - ctz64(m); break; }
	}
	lo = a; hi = bb + 1;
	if (n <= 0) { lo = 0; hi = 0; }
	return a;
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// One read held in registers: lane l holds bases/quals [4l, 4l+4) and [256+4l, 256+4l+4).
struct ReadRegs { unsigned b[2], q[2]; int len; long long start; };

__device__ __forceinline__ void read_prefetch(const AsmArgs &a, long long off, int len, long long start, ReadRegs &R)
{
	const int lane = lane_id();
	R.len = len; R.start = start;
#pragma unroll
	for (int w = 0; w < 2; ++w) {
		const int i = w * 256 + 4 * lane;
		R.b[w] = 0; R.q[w] = 0xffffffffu;
		if (i < len) {
			R.b[w] = *(const u32_unaligned *)(a.bases + off + i);
			if (a.quals) R.q[w] = *(const u32_unaligned *)(a.quals + off + i);
		}
	}
}

// indelope.nim:23-38 on a read held in registers (len <= 512).  Returns a; kept range [lo, hi).
__device__ __forceinline__ int read_trim_regs(const ReadRegs &R, int min_quality, int &lo, int &hi)
{
	const int lane = lane_id();
	const int n = R.len, high = n - 1;
	const unsigned mq = (unsigned)min_quality & 0xff;
	if (n <= 0) { lo = 0; hi = 0; return 0; }
	unsigned good[2];                                         // bit j: quality of base w*256+4l+j >= mq
#pragma unroll
	for (int w = 0; w < 2; ++w) {
		good[w] = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) if (((R.q[w] >> (8 * j)) & 0xff) >= mq) good[w] |= 1u << j;
	}
	int a = high;                                             // :25 first i < high with q[i] >= mq, else high
#pragma unroll
	for (int w = 1; w >= 0; --w) {
		const int base = w * 256 + 4 * lane;
		unsigned m4 = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) if (base + j < high && (good[w] >> j & 1)) m4 |= 1u << j;
		const unsigned long long bm = ballot(m4 != 0);
		if (bm) { const int L = ctz64(bm); a = w * 256 + 4 * L + (__ffs((int)bcast((int)m4, L)) - 1); }
	}
	if (a == high) { lo = 0; hi = 0; return a; }              // :28-30
	int bb = a;                                               // :33 last i > a with q[i] >= mq, else a
#pragma unroll
	for (int w = 0; w < 2; ++w) {
		const int base = w * 256 + 4 * lane;
		unsigned m4 = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) if (base + j > a && base + j <= high && (good[w] >> j & 1)) m4 |= 1u << j;
		const unsigned long long bm = ballot(m4 != 0);
		if (bm) { const int L = 63 - clz64(bm); bb = w * 256 + 4 * L + (31 - __clz((int)bcast((int)m4, L))); }
	}
	lo = a; hi = bb + 1;
	return a;
}

// Trimmed read [lo,hi) from registers into the LDS staging area.
__device__ __forceinline__ void read_stage_regs(const ReadRegs &R, Arena &A, int lo, int hi)
{
	const int lane = lane_id();
#pragma unroll
	for (int w = 0; w < 2; ++w) {
		const int base = w * 256 + 4 * lane;
		if (base >= hi || base + 4 <= lo) continue;
		if (lo == 0) *(uint32_t *)(A.seq + A.stage_off + base) = R.b[w];      // staging is 4-byte aligned
		else {
#pragma unroll
			for (int j = 0; j < 4; ++j)
				if (base + j >= lo && base + j < hi) A.seq[A.stage_off + base + j - lo] = (uint8_t)(R.b[w] >> (8 * j));
		}
	}
}

// assemble (indelope.nim:157-183) for one region.  FAST = max_mismatch == 0: read phase with the
// exact scan and difference-array supports (contig_dev.h "Fast paths"); otherwise the generic scan.
// The next read's bases and qualities are fetched from HBM into registers while the current one is
// scanned, so the read loop never waits on global memory.
template <class ST>
__device__ inline int assemble_region(const AsmArgs &a, ST &S, Arena &A, int r, int &n_pre, int &n_final)
{
	constexpr int QSLOT = ST::QSLOT;
	const int lane = lane_id();
	const long long r0 = a.region_read_off[r], r1 = a.region_read_off[r + 1];
	const bool fast = a.max_mismatch == 0;
	for (int i = lane; i <= ST::MAXC; i += 64) S.alive[i] = 0;
	if (lane == 0) { S.bump = 0; S.err = 0; }
	WSYNC();
	int n = 0;
	n_pre = 0; n_final = 0;
	for (long long g0 = r0; g0 < r1; g0 += 64) {               // indelope.nim:163-169, 64 reads of metadata at a time
		const long long my = g0 + lane;
		const bool in = my < r1;
		long long moff = 0, mstart = 0; int mlen = 0, mtlo = 0, mthi = 0; bool ok = false;
		if (in) {
			moff = a.read_off[my]; mlen = (int)(a.read_off[my + 1] - moff); mstart = a.read_start[my];
			if (a.trim_lo) {                                   // clamped to the read
				mtlo = a.trim_lo[my]; mthi = a.trim_hi[my];
				mtlo = mtlo < 0 ? 0 : mtlo > mlen ? mlen : mtlo;
				mthi = mthi > mlen ? mlen : mthi; mthi = mthi < mtlo ? mtlo : mthi;
			}
			ok = a.mapq[my] >= a.min_mapq_assemble && !(a.read_skip && a.read_skip[my]);   // :164-165
		}
		unsigned long long elig = ballot(ok);
		ReadRegs cur, nxt;
		cur.len = 0; nxt.len = 0;
		if (elig) {
			const int k = ctz64(elig);
			const long long off = ((long long)bcast((int)(moff >> 32), k) << 32) | (unsigned)bcast((int)moff, k);
			const long long st = ((long long)bcast((int)(mstart >> 32), k) << 32) | (unsigned)bcast((int)mstart, k);
			read_prefetch(a, off, bcast(mlen, k) <= 512 ? bcast(mlen, k) : 0, st, cur);
		}
		while (elig) {
			const int k = ctz64(elig);
			elig &= elig - 1;
			const long long b0 = ((long long)bcast((int)(moff >> 32), k) << 32) | (unsigned)bcast((int)moff, k);
			const int len = bcast(mlen, k);
			const long long rstart = ((long long)bcast((int)(mstart >> 32), k) << 32) | (unsigned)bcast((int)mstart, k);
			if (elig) {                                        // start fetching the next eligible read now
				const int k2 = ctz64(elig);
				const long long off2 = ((long long)bcast((int)(moff >> 32), k2) << 32) | (unsigned)bcast((int)moff, k2);
				const long long st2 = ((long long)bcast((int)(mstart >> 32), k2) << 32) | (unsigned)bcast((int)mstart, k2);
				const int len2 = bcast(mlen, k2);
				read_prefetch(a, off2, len2 <= 512 ? len2 : 0, st2, nxt);
			}
			int lo = 0, hi = len, o = 0;
			if (a.trim_lo) { lo = bcast(mtlo, k); hi = bcast(mthi, k); o = lo; }   // :168 done by the stager
			else if (len <= 512) {
				if (a.quals) o = read_trim_regs(cur, a.trim_min_qual, lo, hi);    // :168
				else if (len == 1) hi = 0;                 // no qualities = all 255: trim() still empties a 1-base read (:28-30)
			} else if (a.quals) o = read_trim_dev(a.quals + b0, len, a.trim_min_qual, lo, hi);
			const int tl = hi - lo;
			if (tl > a.stage_cap || tl > MAXLEN) return IHP_E_CAPACITY;
			if (len <= 512) read_stage_regs(cur, A, lo, hi);
			else for (int i = lane; i < tl; i += 64) A.seq[A.stage_off + i] = a.bases[b0 + lo + i];
			if (!fast) for (int i = lane; i < tl; i += 64) A.sup[A.stage_off + i] = 1u;
			if (lane == 0) {                                   // make_contig, contig.nim:143-150
				S.off[QSLOT] = A.stage_off; S.len[QSLOT] = tl; S.cap[QSLOT] = tl;
				S.nreads[QSLOT] = 1; S.start[QSLOT] = rstart + o;
				S.smin[QSLOT] = 1; S.smax[QSLOT] = 1; S.lo3[QSLOT] = 0x3fffffff; S.hi3[QSLOT] = 0;
			}
			if (fast) LDS_ORDER(); else WSYNC();
			const int min_overlap = (int)(a.min_overlap_pct * (double)tl);   // :169
			if (fast) {
				Best b;
				{ IHP_T0(A); b = best_match_read(S, A, S.listA, n, min_overlap); IHP_T1(A, 12); }   // contig.nim:243-244
				IHP_T0(A);
				if (b.found) {
					const int rc = insert_read(S, A, b.slot, b.off);     // contig.nim:246
					if (rc) return rc;
				} else {                                       // contig.nim:248
					if (n >= ST::MAXC) return IHP_E_CAPACITY;
					int slot;
					const int rc = new_contig_from_read(S, A, slot);
					if (rc) return rc;
					if (lane == 0) S.listA[n] = (short)slot;
					n++;
				}
				IHP_T1(A, 13);
				LDS_ORDER();
			} else {
				Best b = best_match_dev(S, A, QSLOT, S.listA, n, min_overlap, a.max_mismatch);
				if (b.found) {
					const int nc = emit_corrections(S, A, QSLOT, b.slot, b.off, IHP_ALLOW_DEFAULT);
					if (nc < 0) return IHP_E_CAPACITY;
					const int rc = insert_dev(S, A, b.slot, QSLOT, b.off, nc);
					if (rc) return rc;
				} else {
					const int slot = alloc_slot(S);
					if (slot < 0 || n >= ST::MAXC) return IHP_E_CAPACITY;
					int need = align4(tl + headroom(tl));
					if (!ensure_space2(S, A, need, false)) { need = align4(tl); if (!ensure_space2(S, A, need, false)) return IHP_E_CAPACITY; }
					const int noff = S.bump;
					for (int i = lane; i < tl; i += 64) { A.seq[noff + i] = A.seq[A.stage_off + i]; A.sup[noff + i] = 1u; }
					if (lane == 0) {
						S.off[slot] = noff; S.len[slot] = tl; S.cap[slot] = need; S.nreads[slot] = 1;
						S.start[slot] = S.start[QSLOT]; S.alive[slot] = 1; S.bump = noff + need + SLOT_PAD;
						// support 1 everywhere, no ">= 3" run: slide_scan reads these of every target (a slot's old values, or
						// whatever the LDS held, sent its 8-base window anywhere -- a memory fault in the HBM-arena pass)
						S.smin[slot] = 1; S.smax[slot] = 1; S.lo3[slot] = 0x3fffffff; S.hi3[slot] = 0;
						S.listA[n] = (short)slot;
					}
					n++;
				}
				WSYNC();
			}
			cur = nxt;
		}
	}
	WSYNC();
	n_pre = n;                                                 // :171
	const long long tcA = a.prof ? (long long)clock64() : 0;
	{ IHP_T0(A); if (fast) materialize_supports(S, A, S.listA, n); IHP_T1(A, 14); }
	if (!fast) for (int i = 0; i < n; ++i) recompute_minmax(S, A, S.listA[i]);
	// combine(min_support) = pass with min_support 0, then the trimmed pass (contig.nim:259-260)
	const int n2 = combine_pass(S, A, S.listA, n, S.listB, 0, a.combine_min_overlap, a.max_mismatch);
	if (n2 < 0) return n2;
	WSYNC();
	const int n3 = combine_pass(S, A, S.listB, n2, S.listA, a.combine_min_support, a.combine_min_overlap, a.max_mismatch);
	if (n3 < 0) return n3;
	WSYNC();
	n_final = n3;
	if (a.prof && lane == 0) S.prof[1] += (long long)clock64() - tcA;
	return 0;
}

// Final contigs of one region -> output slots, max_stop, the faidx-style window of every contig that passes
// indelope.nim:209-211 and its alignment job (:213-220).
template <class ST>
__device__ inline void region_epilogue(const AsmArgs &a, ST &S, Arena &A, int r, int err, int n_pre, int &n_final)
{
	const int lane = lane_id();
	if (err) n_final = 0;
	const long long r0 = a.region_read_off[r], r1 = a.region_read_off[r + 1];
	// max_stop over reads with mapq > 5 (indelope.nim:213-216)
	long long mstop = -0x7fffffffffffffffll - 1;
	for (long long ri = r0 + lane; ri < r1; ri += 64)
		if (a.mapq[ri] > a.min_mapq_stop && a.read_stop[ri] > mstop) mstop = a.read_stop[ri];
	mstop = -wave_min_ll(-mstop - 1) - 1;                  // wave max without negating LLONG_MIN
	const long long seq_base = r0 < r1 ? a.read_off[r0] : 0;
	const long long origin = a.ref_origin[r];
	const long long L = a.ref_off[r + 1] - a.ref_off[r];
	const int width = (int)((double)(a.K + 1) / 2.0 - 1.0);   // :218
	long long cursor = 0;
	for (int k = 0; k < n_final; ++k) {
		const int c = S.listA[k];
		const long long slot = r0 + k;
		const int len = S.len[c];
		const uint8_t *cs = A.seq + S.off[c]; const uint32_t *cp = A.sup + S.off[c];
		// a contig of one read has support 1 on every base (contig.nim:143-150; nothing was ever merged into it): the
		// supports in HBM need not be read back, which would be one more round trip in the region's chain
		if (S.nreads[c] == 1) { for (int i = lane; i < len; i += 64) { a.out_seq[seq_base + cursor + i] = cs[i]; a.out_sup[seq_base + cursor + i] = 1u; } }
		else for (int i = lane; i < len; i += 64) { a.out_seq[seq_base + cursor + i] = cs[i]; a.out_sup[seq_base + cursor + i] = cp[i]; }
		if (lane == 0) {
			const long long cstart = S.start[c], cn = S.nreads[c];
			a.ctg_start[slot] = cstart; a.ctg_nreads[slot] = cn; a.ctg_len[slot] = len;
			a.ctg_seq_off[slot] = seq_base + cursor;
			int flags = 0; long long rs = 0; int rl = 0;
			if (n_pre <= a.max_pre_contigs && cn >= a.min_reads && len >= a.min_ctg_len) {   // :209-211
				const long long max_stop = cstart > mstop ? cstart : mstop;
				// fai.get(chrom, ctg.start, max_stop+width+50) :220 -- faidx_fetch_seq clamping
				long long beg = cstart - origin, end = max_stop + width + a.ref_pad - origin;
				int clamped = 0;
				if (end < beg) { beg = end; clamped = 1; }
				if (beg < 0) { beg = 0; clamped = 1; } else if (L <= beg) { beg = L - 1; clamped = 1; }
				if (end < 0) { end = 0; clamped = 1; } else if (L <= end) { end = L - 1; clamped = 1; }
				long long reflen = L > 0 ? end - beg + 1 : 0;
				if (L <= 0) { beg = 0; clamped = 1; }
				flags = IHP_ALN_DONE | (clamped ? IHP_ALN_REF_CLAMPED : 0);
				rs = origin + beg; rl = (int)reflen;
				const int j = atomicAdd(a.n_jobs, 1);
				AlnJob jb;
				jb.q_off = seq_base + cursor; jb.t_off = a.ref_off[r] + beg; jb.qlen = len; jb.tlen = rl;
				jb.out = (int)slot; jb.region = r; jb.flags = 0; jb.pad_ = 0;
				a.jobs[j] = jb;
			}
			a.aln_flags[slot] = flags; a.aln_ref_start[slot] = rs; a.aln_ref_len[slot] = rl;
		}
		cursor += len;
	}
	if (lane == 0) { a.status[r] = err; a.n_pre[r] = n_pre; a.n_final[r] = n_final; }
}

// MC contig slots; LDSA: contig bases in an LDS arena of a.lds_arena bytes (dynamic LDS).  Three passes
// share this kernel: <64,true> with a small arena for typical regions at 16 waves/CU, <128,true> with a
// large arena for regions whose reads cannot fit the small one, <1024,false> with an HBM arena as the
// catch-all.  A region is forwarded (out_list) either up front, from its read bases, or when it runs out
// of arena / contig slots; results never depend on which pass produced them.
template <int MC, bool LDSA, int MINW>
__global__ __launch_bounds__(64, MINW) void k_assemble(const AsmArgs a)
{
	typedef RegionStateT<MC> ST;
	__shared__ ST S;
	__shared__ int s_item;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_arena[];
	const int lane = lane_id();
	Arena A;
	A.sup = a.arena_sup + (size_t)blockIdx.x * a.arena_cap;
	if (LDSA) { A.seq = lds_arena; A.cap = a.lds_arena - 16; A.stage_off = a.lds_arena - 16 - a.stage_cap; }
	else { A.seq = a.arena_seq + (size_t)blockIdx.x * a.arena_cap; A.cap = a.arena_cap - 16; A.stage_off = a.arena_cap - 16 - a.stage_cap; }
	A.corr = a.corr + (size_t)blockIdx.x * a.corr_cap; A.corr_cap = a.corr_cap; A.prof = a.prof ? S.prof : nullptr;
	mark_start(a.t_start);
	if (lane < 16) S.prof[lane] = 0;
	WSYNC();
	const int n_items = a.in_list ? *a.n_in : a.n_regions;
	unsigned wq_dead = 0;
	for (;;) {
		if (lane == 0) s_item = wq_next(a.work_counter, n_items, (int)blockIdx.x, wq_dead);
		WSYNC();
		int r = __builtin_amdgcn_readfirstlane(s_item);
		WSYNC();
		if (r < 0) break;
		if (a.in_list) r = a.in_list[r];
		if (LDSA && a.out_list) {
			// live contig bytes (with headroom) stay below ~30% of the read bases on indel-region pile-ups;
			// do not start what is unlikely to fit (a wrong guess only costs the forward on overflow)
			const long long nb = a.read_off[a.region_read_off[r + 1]] - a.read_off[a.region_read_off[r]];
			if (nb * 3 / 10 + 2 * a.stage_cap > a.lds_arena) {
				if (lane == 0) a.out_list[atomicAdd(a.n_out, 1)] = r;
				WSYNC();
				continue;
			}
		}
		int n_pre = 0, n_final = 0;
		const long long tcR = a.prof ? (long long)clock64() : 0;
		int err = assemble_region(a, S, A, r, n_pre, n_final);
		WSYNC();
		if (a.prof && lane == 0) { S.prof[0] += (long long)clock64() - tcR; S.prof[3] += 1; }
		if (err == IHP_E_CAPACITY && a.out_list) {             // out of arena / contig slots: next pass
			if (lane == 0) a.out_list[atomicAdd(a.n_out, 1)] = r;
			WSYNC();
			continue;
		}
		region_epilogue(a, S, A, r, err, n_pre, n_final);
		if (a.prof && lane == 0) S.prof[2] += (long long)clock64() - tcR;
		WSYNC();
	}
	WSYNC();
	if (a.prof && lane < 16 && lane != 8 && lane != 9 && lane != 10 && lane != 11 && S.prof[lane])
		atomicAdd((unsigned long long *)&a.prof[lane], (unsigned long long)S.prof[lane]);
}


// Class 1 with the packed read phase (asm2_dev.h), as two kernels so that each runs at the occupancy its own state allows:
//   k_asm_reads     the read insertions on 2-bit bases: registers + a small packed area in LDS, 32 waves per CU; leaves one
//                   hand-over record per region in HBM (contig directory, read records, packed bases);
//   k_asm_combine3  takes a record over (packed bases as they are, supports counted from the records into one byte per
//                   base, both in LDS: asm3_dev.h) and runs combine (contig.nim:254-281) and the epilogue on it.
// A region that does not meet the packed path's preconditions, or runs out of room in either kernel, goes to out_list and
// is assembled from scratch by the byte-based passes (k_assemble); results never depend on the pass.
template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_asm_reads(const ReadArgs a)
{
	__shared__ int s_item;
	__shared__ long long s_prof[16];
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_arena[];
	uint32_t *P = (uint32_t *)lds_arena;
	const int lane = lane_id();
	mark_start(a.t_start);
	if (lane < 16) s_prof[lane] = 0;
	WSYNC();
	const int n_items = a.in_list ? *a.n_in : a.n_regions;
	unsigned wq_dead = 0;
	for (;;) {
		if (lane == 0) s_item = wq_next(a.work_counter, n_items, (int)blockIdx.x, wq_dead);
		WSYNC();
		int r = __builtin_amdgcn_readfirstlane(s_item);
		WSYNC();
		if (r < 0) break;
		if (a.in_list) r = a.in_list[r];
		int nc = 0, need = 0;
		const bool deep = uni(a.region_read_off[r + 1]) - uni(a.region_read_off[r]) > 256;     // (what the byte build keeps records for; 256 reads on ONE base hand a region to the retry route as before)
		const int err = v2_read_phase(a, P, a.v2_pdw, r, a.prof ? s_prof : nullptr, nc, need);
		if (err) {                                             // not for this path: the byte-based passes take it
			// (n_final = 0: the launches that take the list may have been left out of this run -- ihp_batch_run -- and k_summary walks n_final contigs)
			if (lane == 0) { a.v2_hand[a.v2_hoff[r]] = 0xffffffffu; a.n_final[r] = 0; a.out_list[atomicAdd(a.n_out, 1)] = r; }
		} else if (lane == 0 && a.hist && !deep) {
			// (a region with more contigs than the first tier's table holds counts as one that only the roomiest first tier would take)
			if (nc > a.manyc_thr) atomicAdd(a.n_manyc, 1);
			if (nc > a.tier_a_maxc) atomicAdd(&a.hist[10], 1);
			else for (int k = 0; k < 11; ++k) if (need <= a.hist_cap[k]) { atomicAdd(&a.hist[k], 1); break; }
		}
		if (!err && a.lpt_cnt && lane == 0) {
			// the tiers in memory: first, third, second (the first tier's launch can then walk the third's lists, or all, behind its own)
			// (the fourth tier: the regions of more than 256 reads, whose supports need not fit a byte -- the wide build's launch)
			const int tier = deep ? 3 : need <= a.tier_a_cap && nc <= a.tier_a_maxc ? 0 : need <= a.tier_b_cap ? 2 : 1;
			const int c = lpt_class(nc) + tier * LPT_CLASSES;
			a.lpt_seg[(size_t)c * a.lpt_stride + atomicAdd(&a.lpt_cnt[c], 1)] = r;
			if (tier == 1 || tier == 2) atomicAdd(a.n_tier_b + (tier == 2 ? 0 : 1), 1);
			if (a.prof) {                                          // what the regions ask of the combine arena: sum and maximum of the capacity units
				atomicAdd((unsigned long long *)&a.prof[62], (unsigned long long)need);
				atomicMax((unsigned long long *)&a.prof[63], (unsigned long long)need);
			}
		}
		WSYNC();
	}
	WSYNC();
	if (a.prof && lane < 16 && s_prof[lane]) atomicAdd((unsigned long long *)&a.prof[lane == 7 ? 27 : lane], (unsigned long long)s_prof[lane]);
}

// Final contigs of one region from the packed representation (asm3_dev.h) -> output slots, alignment jobs; see region_epilogue.
template <class ST>
__device__ inline void region_epilogue3(const AsmArgs &a, ST &S, const V3Ctx &C, int r, int err, int n_pre, int &n_final, V3Next &N)
{
	const int lane = lane_id();
	if (err) n_final = 0;
	v3n_ticket(N);                                                   // (the wave's next region: asm3_dev.h, V3Next)
	const long long r0 = a.region_read_off[r], r1 = a.region_read_off[r + 1];
	long long mstop = -0x7fffffffffffffffll - 1;                   // max_stop over reads with mapq > 5 (indelope.nim:213-216)
	if (r0 < r1) v3n_region(a, N);                                   // (r0 is here, and so is the ticket)
	for (long long ri = r0 + lane; ri < r1; ri += 64) {                // (both loads at once: one round trip a step, not two)
		const int mq = a.mapq[ri]; const long long rs = a.read_stop[ri];
		mstop = ((mq > a.min_mapq_stop) & (rs > mstop)) ? rs : mstop;
	}
	mstop = -wave_min_ll(-mstop - 1) - 1;
	if (N.stage == 2) v3n_offset(a, N);
	const long long seq_base = r0 < r1 ? a.read_off[r0] : 0;
	const long long origin = a.ref_origin[r];
	const long long roff = a.ref_off[r], L = a.ref_off[r + 1] - roff;
	const int width = (int)((double)(a.K + 1) / 2.0 - 1.0);       // :218
	// The contigs' records and alignment jobs, lane k <-> final contig k (at most 64): one atomic per region for the job slots
	// (per job it was a dependent round trip to L2 in the one lane that did this, and 200 000 requests to one address).
	// In front of the contigs' bases: the atomic's answer is then not queued behind a region's worth of stores.
	long long carry = 0;                                                // bases of the contigs before k0
	for (int k0 = 0; k0 < n_final; k0 += 64) {
		const int k = k0 + lane;
		const bool mine = k < n_final;
		const int c = mine ? (int)S.listA[k] : 0;
		const int len = mine ? S.len[c] : 0;
		const long long mycur = carry + (long long)(wave_scan_add((unsigned)len) - (unsigned)len);
		carry += wave_sum_i(len);
		const long long slot = r0 + k;
		int flags = 0; long long rs = 0, beg = 0; int rl = 0;
		bool job = false;
		if (mine) {
			const long long cstart = S.start[c], cn = S.nreads[c];
			a.ctg_start[slot] = cstart; a.ctg_nreads[slot] = cn; a.ctg_len[slot] = len;
			a.ctg_seq_off[slot] = seq_base + mycur;
			if (n_pre <= a.max_pre_contigs && cn >= a.min_reads && len >= a.min_ctg_len) {   // :209-211
				const long long max_stop = cstart > mstop ? cstart : mstop;
				long long end = max_stop + width + a.ref_pad - origin;      // :220, faidx clamping
				beg = cstart - origin;
				int clamped = 0;
				if (end < beg) { beg = end; clamped = 1; }
				if (beg < 0) { beg = 0; clamped = 1; } else if (L <= beg) { beg = L - 1; clamped = 1; }
				if (end < 0) { end = 0; clamped = 1; } else if (L <= end) { end = L - 1; clamped = 1; }
				const long long reflen = L > 0 ? end - beg + 1 : 0;
				if (L <= 0) { beg = 0; clamped = 1; }
				flags = IHP_ALN_DONE | (clamped ? IHP_ALN_REF_CLAMPED : 0);
				rs = origin + beg; rl = (int)reflen;
				job = true;
			}
			a.aln_flags[slot] = flags; a.aln_ref_start[slot] = rs; a.aln_ref_len[slot] = rl;
		}
		const int j = plan_append(a.n_jobs, job);
		if (job) {
			AlnJob jb;
			jb.q_off = seq_base + mycur; jb.t_off = roff + beg; jb.qlen = len; jb.tlen = rl;
			jb.out = (int)slot; jb.region = r; jb.flags = ALN_Q_ACGT; jb.pad_ = 0;   // 2-bit packed contigs; enc_base() folds the window
			a.jobs[j] = jb;
		}
	}
	// the bases and supports of the final contigs (stores only: nothing below waits for them)
	long long cursor = 0;
	for (int k = 0; k < n_final; ++k) {
		const int c = uni((int)S.listA[k]);
		const int len = uni(S.len[c]), pb = uni(16 * S.dw[c] + S.sh[c]), so = uni(S.so[c]);
		uint8_t *oseq = a.out_seq + seq_base + cursor; uint32_t *osup = a.out_sup + seq_base + cursor;
		for (int i = 4 * lane; i < len; i += 256) {                  // four bases per lane: 2-bit codes -> "ACTG" bytes
			const int b = pb + i;
			const unsigned c8 = (fsh(C.PM[(b >> 4) + 1], C.PM[b >> 4], 2u * (unsigned)(b & 15))) & 0xffu;
			const unsigned t = c8 | (c8 << 6), u = t | (t << 12);
			const unsigned w = __builtin_amdgcn_perm(0u, PK_LUT, u & 0x03030303u);
			unsigned s4[4] = {1u, 1u, 1u, 1u};                           // four supports (the slot is padded: reading past len is fine)
			if (so >= 0) {
				if (ST::WIDE) {
					const unsigned lo = ld32u((const uint32_t *)C.SUP, 2 * (so + i)), hi = ld32u((const uint32_t *)C.SUP, 2 * (so + i) + 4);
					s4[0] = lo & 0xffffu; s4[1] = lo >> 16; s4[2] = hi & 0xffffu; s4[3] = hi >> 16;
				} else {
					const unsigned sv = ld32u((const uint32_t *)C.SUP, so + i);
					s4[0] = sv & 0xffu; s4[1] = (sv >> 8) & 0xffu; s4[2] = (sv >> 16) & 0xffu; s4[3] = sv >> 24;
				}
			}
			if (i + 4 <= len) {
				*(u32_unaligned *)(oseq + i) = w;
				osup[i] = s4[0]; osup[i + 1] = s4[1]; osup[i + 2] = s4[2]; osup[i + 3] = s4[3];
			} else {
#pragma unroll
				for (int j = 0; j < 3; ++j) if (i + j < len) { oseq[i + j] = (uint8_t)(w >> (8 * j)); osup[i + j] = s4[j]; }
			}
		}
		cursor += len;
	}
	if (lane == 0) { a.status[r] = err; a.n_pre[r] = n_pre; a.n_final[r] = n_final; }
}

// combine + epilogue on packed bases and u8 supports (asm3_dev.h).  Dynamic LDS: a.lds_arena bytes of supports, then
// a.v2_pm_dw dwords of packed bases.  A region outside this path's preconditions, or one that runs out of room, goes to
// out_list (the next, roomier launch of this kernel, or the byte-based passes).
// Two builds.  TEAM = false: one wave per workgroup, 96 VGPRs (the first tiers: 12-16 workgroups per CU).  TEAM = true:
// workgroups of 2 or 4 waves for the tiers whose arenas leave 8 or fewer workgroups per CU -- wave 0 runs the region, the
// others join it inside best_match calls (V3Par, asm3_dev.h) and wait at a barrier otherwise; so wave 0 itself never uses a
// workgroup barrier outside those calls (WAVE_SYNC: its own memory traffic done, nothing more).  128 VGPRs, 16 waves per CU.
#define WAVE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
// WIDE: 16-bit supports and the records of up to 640 reads (asm3_dev.h): the launch of the regions with more than 255 reads.
template <int MINW, bool TEAM, int MAXC = V3_MAXC, bool WIDE = false>
__global__ __launch_bounds__(TEAM ? 64 * V3_MAXW : 64) __attribute__((amdgpu_waves_per_eu(MINW, 8))) void k_asm_combine3(const AsmArgs a)
{
	__shared__ V3StateT<MAXC, WIDE> S;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds_arena[];
	__shared__ V3Par s_par;                                    // (dropped from the one-wave build: nobody refers to it there)
	V3Par *const par = TEAM ? &s_par : nullptr;
	const int lane = lane_id();
	const int wave = TEAM ? (int)(threadIdx.x >> 6) : 0, n_waves = TEAM ? (int)(blockDim.x >> 6) : 1;
	V3Ctx C;
	C.SUP = lds_arena; C.sup_cap = a.lds_arena; C.PM = (uint32_t *)(lds_arena + (size_t)a.lds_arena * sizeof(typename V3StateT<MAXC, WIDE>::sup_t)); C.pm_cap = a.v2_pm_dw;   // (a.lds_arena: supports, not bytes)
	C.bump_pm = C.bump_sup = 0; C.prof = a.prof ? S.prof : nullptr; C.cnt = a.prof ? S.cnt : nullptr;
	if (TEAM && wave != 0) {
		C.prof = nullptr; C.cnt = nullptr;
		v3_helper_loop(S, C, par, wave, n_waves);
		return;
	}
	V3Team T;
	T.par = par; T.nparts = n_waves; T.list_b = 0; T.n = 0; T.dver = 0;
	mark_start(a.t_start);
	if (lane < 16) S.prof[lane] = 0;
	if (lane < 16) S.cnt[lane] = 0;
	WAVE_SYNC();
	int n_items = a.in_list ? *a.n_in : a.n_regions;
	// The cost classes laid end to end: lane c keeps where class c ends (the counts are final: the read phase's kernel is over).  An
	// item is then a ballot and a v_readlane away from its (class, position) -- the loop over the classes' counts that used to stand
	// here was up to fifteen scalar loads, each behind a branch on the one before: ~9 us of a C2 region's ~120.
	int cls_end = 0;
	if (a.lpt_cnt) {
		cls_end = (int)wave_scan_add(lane < a.lpt_nclass ? (unsigned)a.lpt_cnt[lane] : 0u);
		n_items = __builtin_amdgcn_readlane(cls_end, 63);
	}
	// (the work queue: V3Next, asm3_dev.h -- a region's successor is asked for beside the loads of its epilogue)
	V3Next N;
	N.S = (int)gridDim.x < WQ_SHARDS ? (int)gridDim.x : WQ_SHARDS; N.s = (int)blockIdx.x % N.S;
	N.n_items = n_items; N.cls_end = cls_end; N.tick_v = 0; N.rn_v = -1; N.hn_v = 0;
	N.ctr = (v3_gint_p)(a.work_counter + N.s * 16);
	v3n_ticket(N);
	v3n_finish(a, N);
	int r = N.rn;
	long long hoff = N.hn;
	while (r >= 0) {
		int n_pre = 0, n_final = 0;
		const long long tcR = a.prof ? (long long)clock64() : 0;
		N.stage = 0;
		int err = v3_take_over(a, S, C, hoff, n_pre);            // (1: the read phase did not take this region)
		const long long tcA = a.prof ? (long long)clock64() : 0;
		if (!err) {
			const int n2 = v3_combine_pass<TEAM>(S, C, S.listA, n_pre, S.listB, 0, a.combine_min_overlap, T);
			if (n2 < 0) err = n2;
			else {
				const int n3 = v3_combine_pass<TEAM>(S, C, S.listB, n2, S.listA, a.combine_min_support, a.combine_min_overlap, T);
				if (n3 < 0) err = n3; else n_final = n3;
			}
		}
		if (a.prof && lane == 0 && err != 1) { S.prof[0] += (long long)clock64() - tcR; S.prof[1] += (long long)clock64() - tcA; S.prof[3] += 1; }
		if (err == 1) {
		} else if (err == IHP_E_CAPACITY && a.out_list) {       // not here: the next, roomier launch (or the byte-based passes) take it
			if (lane == 0) { a.n_final[r] = 0; a.out_list[atomicAdd(a.n_out, 1)] = r; }
		} else {
			region_epilogue3(a, S, C, r, err, n_pre, n_final, N);
			if (a.prof && lane == 0) {
				const long long dt_ = (long long)clock64() - tcR;
				S.prof[2] += dt_;
				atomicMax((unsigned long long *)&a.prof[53], ((unsigned long long)dt_ << 20) | (unsigned long long)(r & 0xfffff));   // the longest region of the launch (cycles << 20 | region)
				atomicAdd((unsigned long long *)&a.prof[54 + (n_pre >= 19 ? 0 : n_pre >= 13 ? 1 : 2)], (unsigned long long)dt_);
				atomicAdd((unsigned long long *)&a.prof[57 + (n_pre >= 19 ? 0 : n_pre >= 13 ? 1 : 2)], 1ull);
			}
		}
		if (N.stage == 0) v3n_ticket(N);                          // (a region without an epilogue)
		v3n_finish(a, N);
		r = N.rn; hoff = N.hn;
	}
	if (TEAM && n_waves > 1) {                                 // the others leave their loop
		if (lane == 0) par->cmd = 2;
		__syncthreads();
	}
	WAVE_SYNC();
	if (a.prof && lane < 16 && lane != 8 && lane != 9 && lane != 10 && lane != 11 && S.prof[lane])
		atomicAdd((unsigned long long *)&a.prof[lane], (unsigned long long)S.prof[lane]);
	if (a.prof && lane >= 8 && lane < 12 && S.prof[lane]) atomicAdd((unsigned long long *)&a.prof[40 + lane], (unsigned long long)S.prof[lane]);
	if (a.prof && lane < 16 && S.cnt[lane]) atomicAdd((unsigned long long *)&a.prof[32 + lane], (unsigned long long)S.cnt[lane]);
}

// ---------------------------------------------------------------------- ksw2
constexpr int CIG_SLOT = 32;      // CIGAR words reserved per alignment job (longer ones go to the bump pool)

struct KswArgs {
	const AlnJob *jobs; const int *n_jobs; int n_jobs_host;   // n_jobs may be null (use n_jobs_host)
	const uint8_t *qbase, *tbase;
	KswParams P;
	int lds_budget;
	uint8_t *p_scratch; size_t p_cap;          // per workgroup
	uint32_t *cig_tmp; int cig_cap;            // per workgroup
	KswOut *ez;                                // [slots]
	long long *cig_off;                        // [slots] offset into the pool (or -1)
	uint32_t *cig_pool; unsigned long long *cig_cursor; long long cig_pool_cap;   // [0, cig_bump_cap): bump region,
	long long cig_bump_cap;                    // then one CIG_SLOT-word slot per job
	int *overflow;                             // [0] cigar pool, [1] LDS/p budget
	int *work_counter;
	// A job whose sequences need more LDS or traceback scratch than this launch gives every wave is put on ovf_list (when there
	// is one) and gets an empty record for now; a second launch of the same kernel -- in_list = that list, a few workgroups with
	// all the LDS and a large scratch each -- aligns it.  Without ovf_list such a job raises overflow[1] (IHP_E_CAPACITY).
	const int *in_list;                        // work item -> job (null: item j is job j)
	const int2 *pairs;                         // k_ksw_pair: work item -> two jobs of equal qlen (k_ksw_plan); n_jobs counts pairs
	int *ovf_list, *ovf_n;
	long long *prof;                           // optional cycle counters (diagnostics)
	unsigned long long *t_start;               // optional: see mark_start()
	signed char gmat[64]; int gm;              // KSW_EZ_GENERIC_SC: the m x m score matrix (gm = m <= 8, else 0); MODE 2 only
};

// MODE 3: top-byte register-resident sweep (ksw_narrow.h), left-aligned gaps; 4: same, KSW_EZ_RIGHT -- the
// production kernels.  0/1: the masked register-resident sweep (ksw_fast.h) for scoring schemes or alphabets
// ksw_narrow_ok() rejects; 2: generic LDS sweep (any band).  Separate instantiations keep each kernel's
// register footprint to what its sweep needs.
// The launch arguments are read through the kernarg segment pointer, re-derived behind an empty asm at the top
// of every job and again after the sweep: the ~40 SGPRs of pointers and capacities are then loaded where they are
// used instead of staying live (and being spilled to VGPR lanes) across the sweep's inner loop.
typedef const __attribute__((address_space(4))) KswArgs *KswArgsK;
__device__ __forceinline__ KswArgsK ksw_args_again(KswArgsK p) { asm volatile("" : "+s"(p)); return p; }

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == 3 || MODE == 4 ? 8 : 1))) void k_ksw(const KswArgs)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	__shared__ int s_item;
	__shared__ long long s_off;
	const int lane = lane_id();
	const KswArgsK a0 = (KswArgsK)__builtin_amdgcn_kernarg_segment_ptr();
	{
		const KswArgsK a = a0;
		mark_start(a->t_start);
		if (a->prof && lane < 6) ((long long *)(lds + a->lds_budget + 16))[lane] = 0;   // per-wave cycle counters live past the sweep's LDS
	}
	WSYNC();
	{
		unsigned wq_dead = 0;
		for (;;) {
			KswArgsK a = ksw_args_again(a0);
			const int njobs = a->n_jobs ? *a->n_jobs : a->n_jobs_host;
			if (lane == 0) s_item = wq_next(a->work_counter, njobs, (int)blockIdx.x, wq_dead);
			WSYNC();
			const int j = __builtin_amdgcn_readfirstlane(s_item);   // wave-uniform by construction; tells the compiler so
			WSYNC();
			if (j < 0) break;
			const int jj = a->in_list ? a->in_list[j] : j;
			const AlnJob jb = a->jobs[jj];
			KswOut out;
			{
				uint8_t *p = a->p_scratch + (size_t)blockIdx.x * a->p_cap;
				uint32_t *ct = a->cig_tmp + (size_t)blockIdx.x * a->cig_cap;
				long long *pacc = a->prof ? (long long *)(lds + a->lds_budget + 16) : nullptr;
				const KswParams P = {a->P.m, a->P.sc_mch, a->P.sc_mis, a->P.min_sc, a->P.q, a->P.e, a->P.w, a->P.zdrop, a->P.flag,
				                     a->P.encode_ascii, a->P.codes_ok};
				int w = P.w;
				if (w < 0) w = jb.tlen > jb.qlen ? jb.tlen : jb.qlen;
				int ncol_ = jb.qlen < jb.tlen ? jb.qlen : jb.tlen;
				ncol_ = ((ncol_ < w + 1 ? ncol_ : w + 1) + 15) / 16 + 1;
				const size_t pneed = MODE == 3 || MODE == 4 ? ksw_narrow_p_bytes(jb.qlen, jb.tlen)
				                                            : ((size_t)(jb.qlen + jb.tlen - 1 > 0 ? jb.qlen + jb.tlen - 1 : 0) * ncol_ + 1) * 16;
				// MODE 5: bands wider than the register layouts of MODE 0-4 (w < 0 or w > 62): the ring sweep of ksw_wide.h when
				// the job fits it (3 or 6 slots), the LDS sweep otherwise
				const int wide = MODE != 5 ? 0 : ksw_wide_ok<3>(P, jb.qlen, jb.tlen) ? 3 : ksw_wide_ok<6>(P, jb.qlen, jb.tlen) ? 6 : 0;
				const size_t lneed = MODE == 5 ? (wide == 3 ? ksw_wide_lds_bytes<3>(jb.qlen, jb.tlen) : wide == 6 ? ksw_wide_lds_bytes<6>(jb.qlen, jb.tlen) : ksw_lds_bytes(jb.qlen, jb.tlen))
				                     : MODE >= 3 ? ksw_narrow_lds_bytes(jb.qlen, jb.tlen) : MODE != 2 ? ksw_fast_lds_bytes(jb.qlen, jb.tlen) : ksw_lds_bytes(jb.qlen, jb.tlen);
				const uint8_t *qy = a->qbase + jb.q_off, *tg = a->tbase + jb.t_off;
				const int cig_cap = a->cig_cap;
				if (jb.qlen > 0 && jb.tlen > 0 && (lneed > (size_t)a->lds_budget || pneed > a->p_cap)) {
					out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;
					out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = -1;
					if (a->ovf_list) {                                     // the roomy launch takes it; an empty record until then
						if (lane == 0) a->ovf_list[atomicAdd(a->ovf_n, 1)] = jj;
						out.n_cigar = 0;
					}
				} else if (MODE == 3 || MODE == 4) {
					if (!ksw_wave_narrow<MODE == 4>(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out, pacc))
						out.n_cigar = -1;                                  // a code outside the alphabet: the host never sends those here
				} else if (MODE == 0 || MODE == 1) {
					ksw_wave_fast<MODE == 1>(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out, pacc);
				} else if (MODE == 5) {
					bool done = false;
					if (wide == 3) done = ksw_wave_wide<3, false>(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out);
					else if (wide == 6) done = ksw_wave_wide<6, false>(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out);
					if (!done) {                                           // too wide for the ring, or a code outside the alphabet
						if (ksw_lds_bytes(jb.qlen, jb.tlen) > (size_t)a->lds_budget) {
							out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;
							out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = -1;
						} else ksw_wave(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out, pacc);
					}
				} else {
					signed char gmat[64];
					const int gm = a->gm;
					if (gm) for (int i = 0; i < 64; ++i) gmat[i] = a->gmat[i];
					ksw_wave(qy, jb.qlen, tg, jb.tlen, P, lds, p, ct, cig_cap, out, pacc, gm ? gmat : nullptr);
				}
			}
			a = ksw_args_again(a0);
			const uint32_t *ct = a->cig_tmp + (size_t)blockIdx.x * a->cig_cap;
			long long off = -1;
			if (out.n_cigar > 0) {
				if (out.n_cigar <= CIG_SLOT) off = a->cig_bump_cap + (long long)jj * CIG_SLOT;  // the job's own slot: no atomic
				else {
					if (lane == 0) s_off = (long long)atomicAdd(a->cig_cursor, (unsigned long long)out.n_cigar);
					WSYNC();
					off = s_off;
					if (off + out.n_cigar > a->cig_bump_cap) off = a->cig_pool_cap;     // bump region exhausted
				}
				if (off + out.n_cigar <= a->cig_pool_cap) {
					uint32_t *pool = a->cig_pool;
					for (int i = lane; i < out.n_cigar; i += 64) pool[off + i] = ct[i];
				} else {
					if (lane == 0) atomicExch(&a->overflow[0], 1);
					off = -1;
				}
			} else if (out.n_cigar < 0) {
				if (lane == 0) atomicExch(&a->overflow[1], 1);             // LDS / traceback scratch budget
			}
			if (lane == 0) { a->ez[jb.out] = out; a->cig_off[jb.out] = off; }
			WSYNC();
		}
	}
	WSYNC();
	{
		const KswArgsK a = ksw_args_again(a0);
		if (a->prof && lane < 6)      // [8..11] init, DP, traceback, alignments; [60], [61] early / tail diagonals of the narrow sweep
			atomicAdd((unsigned long long *)&a->prof[lane < 4 ? 8 + lane : 56 + lane], (unsigned long long)((long long *)(lds + a->lds_budget + 16))[lane]);
	}
}

// ------------------------------------------------------- ksw2, two alignments per wavefront (ksw_pair.h)
// The plan: which jobs share a wavefront.  A job the pair sweep can take (ksw_pair_job_ok, a producer that vouches for the
// codes, room in the pair launch's LDS and traceback scratch) is counted under its contig length (k_ksw_plan_count: one
// atomic per job, its rank among the jobs of that length kept in `rank`); k_ksw_plan_place turns the counts into positions
// (every workgroup scans the 4096 counts in LDS: cheaper than a launch in between) and the jobs of rank 2i and 2i + 1 of a
// length write the two halves of pair i of that length.  Everything else -- the odd one of a length included -- goes on the
// singles list, which the single sweep's launch walks through in_list (positions from one atomic per wavefront: a ballot
// ranks the lanes).  Two earlier versions: a counting sort in ONE workgroup (0.6 ms for 200 000 jobs) and pairing through a
// compare-and-swap slot per length (retries grow with the jobs in flight: 190 ms for the same batch).  Who is paired with
// whom differs from run to run; results do not depend on it (tests/test_gpu_round4.py).
struct KswPlanArgs {
	const AlnJob *jobs; const int *n_jobs; int n_jobs_host;
	KswParams P; int pair_on;
	int lds_budget; size_t p_cap;              // of the pair launch
	int *count;                                // [PLAN_KEYS] jobs per contig length; zero when the run starts
	int *rank;                                 // [jobs] scratch
	int2 *pairs; int *n_pairs; int *singles; int *n_singles;
	int *done;                                 // workgroups of k_ksw_plan_count that have finished; zero when the run starts
	int *pbase;                                // [PLAN_KEYS] first pair of every contig length (the last workgroup of the count writes it)
	unsigned long long *t_start;               // optional: see mark_start() (the ksw2 stage begins with its plan)
};

__device__ __forceinline__ int ksw_plan_key(const KswPlanArgs &a, int qlen, int tlen, int flags)
{
	return a.pair_on ? plan_key(a.P, a.lds_budget, a.p_cap, qlen, tlen, flags) : -1;
}

__global__ __launch_bounds__(256) void k_ksw_plan_count(const KswPlanArgs a)
{
	const int tid = (int)threadIdx.x;
	mark_start(a.t_start);
	const int n = a.n_jobs ? *a.n_jobs : a.n_jobs_host;
	const int step = (int)(gridDim.x * blockDim.x);
	for (int j0 = (int)(blockIdx.x * blockDim.x); j0 < n; j0 += step) {      // (whole wavefronts stay together for the ballot)
		const int j = j0 + tid;
		int key = -1;
		if (j < n) key = ksw_plan_key(a, a.jobs[j].qlen, a.jobs[j].tlen, a.jobs[j].flags);
		// one atomic per (wavefront, contig length), all of a wavefront's side by side: the lanes of a length rank themselves
		// behind their first lane's result (a length that a fifth of the batch shares is otherwise one address taking 40 000 requests)
		{
			const int lane = (int)(threadIdx.x & 63);
			bool todo = key >= 0;
			int leader = lane, rk = 0, cnt = 0;
			while (const unsigned long long t = ballot(todo)) {
				const int lead = ctz64(t);
				const int k = __builtin_amdgcn_readlane(key, lead);
				const unsigned long long m = ballot(todo && key == k);
				if (todo && key == k) { leader = lead; rk = popc64(m & ((1ull << lane) - 1)); cnt = popc64(m); todo = false; }
			}
			int base = 0;
			if (key >= 0 && leader == lane) base = atomicAdd(a.count + key, cnt);
			base = __builtin_amdgcn_ds_bpermute(leader << 2, base);
			if (key >= 0) a.rank[j] = base + rk;
		}
		const int ss = plan_append(a.n_singles, j < n && key < 0);
		if (ss >= 0) a.singles[ss] = j;
	}
	// The workgroup that finishes last turns the counts into the first pair of every length (a.pbase) and the number of pairs:
	// the placing launch then needs no LDS -- it used to hold the 16 KB scan in every workgroup, and its workgroups waited for
	// that LDS behind the other chain's assembly (70 us on the critical path of a 5 000-region launch for 10 us of work).
	__shared__ int part[256];
	__shared__ int s_last;
	__threadfence();
	__syncthreads();
	if (tid == 0) s_last = atomicAdd(a.done, 1) == (int)gridDim.x - 1;
	__syncthreads();
	if (!s_last) return;
	__threadfence();
	constexpr int PER = PLAN_KEYS / 256;
	int c[PER], sum = 0;
	for (int i = 0; i < PER; ++i) { c[i] = atomicAdd(a.count + tid * PER + i, 0) >> 1; sum += c[i]; }   // (the other workgroups' atomics: read where they landed)
	part[tid] = sum;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		const int v = tid >= d ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	int run = part[tid] - sum;
	for (int i = 0; i < PER; ++i) { a.pbase[tid * PER + i] = run; run += c[i]; }
	if (tid == 255) *a.n_pairs = part[255];
}

__global__ __launch_bounds__(256) void k_ksw_plan_place(const KswPlanArgs a)
{
	const int tid = (int)threadIdx.x;
	const int n = a.n_jobs ? *a.n_jobs : a.n_jobs_host;
	if ((long long)blockIdx.x * blockDim.x >= n) return;                   // (the grid is sized for the job slots, not the jobs)
	const int *pbase = a.pbase;
	const int step = (int)(gridDim.x * blockDim.x);
	for (int j0 = (int)(blockIdx.x * blockDim.x); j0 < n; j0 += step) {
		const int j = j0 + tid;
		int key = -1;
		if (j < n) key = ksw_plan_key(a, a.jobs[j].qlen, a.jobs[j].tlen, a.jobs[j].flags);
		bool odd_one = false;
		if (key >= 0) {
			const int rk = a.rank[j], cnt = a.count[key];
			odd_one = rk == cnt - 1 && (cnt & 1);
			if (!odd_one) ((int *)(a.pairs + pbase[key] + (rk >> 1)))[rk & 1] = j;
		}
		const int ss = plan_append(a.n_singles, odd_one);
		if (ss >= 0) a.singles[ss] = j;
	}
}

// k_ksw_pair: one wavefront per pair of the plan; the launch arguments are the single sweep's (KswArgs: jobs, sequences,
// result slots, CIGAR pool), `pairs` / n_jobs the plan's list.  a->lds_budget and a->p_cap are the plan's limits.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_ksw_pair(const KswArgs)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	__shared__ int s_item;
	__shared__ long long s_off;
	const int lane = lane_id();
	const KswArgsK a0 = (KswArgsK)__builtin_amdgcn_kernarg_segment_ptr();
	{
		const KswArgsK a = a0;
		mark_start(a->t_start);
		if (a->prof && lane < 6) ((long long *)(lds + a->lds_budget + 16))[lane] = 0;
	}
	WSYNC();
	unsigned wq_dead = 0;
	for (;;) {
		KswArgsK a = ksw_args_again(a0);
		const int nitems = *a->n_jobs;
		if (lane == 0) s_item = wq_next(a->work_counter, nitems, (int)blockIdx.x, wq_dead);
		WSYNC();
		const int it = __builtin_amdgcn_readfirstlane(s_item);
		WSYNC();
		if (it < 0) break;
		PairResult R;
		bool ok;
		{
			const int2 pr = a->pairs[it];
			const int j0 = __builtin_amdgcn_readfirstlane(pr.x), j1 = __builtin_amdgcn_readfirstlane(pr.y);
			const AlnJob jb0 = a->jobs[j0], jb1 = a->jobs[j1];
			uint8_t *p = a->p_scratch + (size_t)blockIdx.x * a->p_cap;
			long long *pacc = a->prof ? (long long *)(lds + a->lds_budget + 16) : nullptr;
			ok = ksw_pair_sweep(a->qbase + jb0.q_off, a->tbase + jb0.t_off, jb0.tlen, a->qbase + jb1.q_off, a->tbase + jb1.t_off, jb1.tlen, jb0.qlen,
			                    a->P.w, a->P.q, a->P.e, a->P.sc_mch, a->P.sc_mis, a->P.zdrop, a->P.encode_ascii, lds, p, R, pacc);
		}
		// the tracebacks: the jobs are read again (nothing of them stayed in a register across the sweep)
		for (int k = 0; k < 2; ++k) {
			const KswArgsK b = ksw_args_again(a0);
			const int2 pr = b->pairs[it];
			const int jj = __builtin_amdgcn_readfirstlane(k ? pr.y : pr.x);
			const int qlen = uni(b->jobs[jj].qlen), tlen = uni(b->jobs[jj].tlen), slot = uni(b->jobs[jj].out);
			const uint8_t *p = b->p_scratch + (size_t)blockIdx.x * b->p_cap;
			uint32_t *ct = b->cig_tmp + (size_t)blockIdx.x * b->cig_cap;
			long long *pacc = b->prof ? (long long *)(lds + b->lds_budget + 16) : nullptr;
			KswOut out;
			if (!ok) {                                                     // a code the producer vouched against: refuse loudly
				out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;
				out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = -1;
			} else if (k == 0) ksw_pair_finish<0>(R, p, qlen, tlen, b->P.w, b->P.flag, ct, b->cig_cap, out, pacc);
			else ksw_pair_finish<1>(R, p, qlen, tlen, b->P.w, b->P.flag, ct, b->cig_cap, out, pacc);
			long long off = -1;
			if (out.n_cigar > 0) {
				if (out.n_cigar <= CIG_SLOT) off = b->cig_bump_cap + (long long)jj * CIG_SLOT;   // the job's own slot: no atomic
				else {
					if (lane == 0) s_off = (long long)atomicAdd(b->cig_cursor, (unsigned long long)out.n_cigar);
					WSYNC();
					off = s_off;
					if (off + out.n_cigar > b->cig_bump_cap) off = b->cig_pool_cap;        // bump region exhausted
				}
				if (off + out.n_cigar <= b->cig_pool_cap) {
					uint32_t *pool = b->cig_pool;
					for (int i = lane; i < out.n_cigar; i += 64) pool[off + i] = ct[i];
				} else {
					if (lane == 0) atomicExch(&b->overflow[0], 1);
					off = -1;
				}
			} else if (out.n_cigar < 0) {
				if (lane == 0) atomicExch(&b->overflow[1], 1);
			}
			if (lane == 0) { b->ez[slot] = out; b->cig_off[slot] = off; }
			WSYNC();
		}
	}
	WSYNC();
	{
		const KswArgsK a = ksw_args_again(a0);
		if (a->prof && lane < 6)
			atomicAdd((unsigned long long *)&a->prof[lane < 4 ? 8 + lane : 56 + lane], (unsigned long long)((long long *)(lds + a->lds_budget + 16))[lane]);
	}
}

// --------------------------------------------------------------------- tally
struct TallyArgs {
	const AlnJob *jobs; const int *n_jobs;
	const uint8_t *out_seq, *ref_bases, *bases, *mapq;
	const uint32_t *pk; const uint8_t *read_bad;   // k_prepack's 2-bit bases and per-read "not all upper-case ACGT" flags, or null
	const long long *read_off, *region_read_off, *ref_origin, *ctg_start;
	const KswOut *ez; const long long *cig_off; const uint32_t *cig_pool;
	TallyParams P;
	DevEvent *ev_pool; unsigned long long *ev_cursor; long long ev_pool_cap;
	long long *ev_off; int *n_ev;              // [slots]
	int *overflow;                             // [2] event pool
	int *work_counter;
	long long *prof;                           // optional cycle counters (diagnostics)
	int lds_bytes;                             // dynamic LDS for staging 64 reads
	FbItem *fb_items; int *fb_count;           // events handed to the alignment fallback (one slot per event)
	int *hit_pool; unsigned long long *hit_cursor; long long hit_cap; int *hit_overflow;   // first-hit positions per (tallied event, read)
	int *hit_region_cnt;                       // [R] events of the region that took one of its own hit slots
	long long hit_bump0;                       // start of the shared bump region
	unsigned long long *t_start;               // optional: see mark_start()
	// k_tally_prep -> k_tally: one record per job that has events to tally (see TallyRec), in the order the threads got there
	struct TallyRec *recs; int rec_cap; int *n_recs;   // n_recs[0]: records written (may pass rec_cap), n_recs[1]: jobs on ovf_jobs
	int *ovf_jobs;                             // jobs that found the record array full: k_tally works their header out itself
	int njobs_cap;
};

// What k_tally needs to know of a job before it touches the sequences, gathered by k_tally_prep (one THREAD per job) into one
// 128-byte line.  The header of a job used to be four dependent round trips to HBM (job -> alignment record, CIGAR offset,
// contig start, region bounds -> CIGAR words, read offsets -> "not ACGT" flags of the region's reads) by a wave with nothing else
// to do -- 46 % of the kernel's wave cycles -- and half of the jobs (contigs that equal the reference: no event) were nothing but
// header.  A thread per job hides those trips behind thousands of others; k_tally gets the jobs WITH events only, one load each.
struct TallyRec {
	long long q_off, t_off;                    // dwords 0..3
	long long rr0, rr1;                        // 4..7: the region's reads
	long long base0, end0;                     // 8..11: read_off of the region's first read and of the end of its first group of 64
	long long coff;                            // 12..13: CIGAR words in cig_pool
	int qlen, tlen;                            // 14, 15
	int region, out;                           // 16, 17
	int n_cigar, job;                          // 18, 19
	int ctg_rel;                               // 20: ctg.start - region origin
	int nev, ntrunc;                           // 21, 22: count_events()
	int packed;                                // 23: the 2-bit reads serve (no read of the region has a base that is not upper-case ACGT)
	unsigned cig[8];                           // 24..31: the CIGAR when it has at most eight words
};
static_assert(sizeof(TallyRec) == 128, "TallyRec is one 128-byte line");

// The header of job j.  False: nothing to tally (no event, or more than max_events, indelope.nim:229) -- ev_off / n_ev written.
// Uniform or per-thread: only plain loads.
__device__ inline bool tally_make_rec(const TallyArgs &a, int j, TallyRec &R)
{
	const AlnJob jb = a.jobs[j];
	const KswOut ez = a.ez[jb.out];
	const long long coff = a.cig_off[jb.out];
	int nev = 0, ntrunc = 0;
	if (ez.n_cigar > 0 && coff >= 0) {
		const uint32_t *cg = a.cig_pool + coff;
		const uint32_t max_off = (uint32_t)ez.max_q;
		uint32_t off = 0;
		for (int i = 0; i < ez.n_cigar; ++i) {                 // count_events (ksw2.nim:22-33)
			if (off >= max_off) break;
			const uint32_t w = cg[i], op = w & 0xf, len = w >> 4;
			if (i < 8) R.cig[i] = w;
			if (op != 2) off += len;
			ntrunc++;
			if (op == 1 || op == 2) nev++;
		}
		for (int i = ntrunc; i < 8 && i < ez.n_cigar; ++i) R.cig[i] = cg[i];
	}
	if (!(nev > 0 && nev <= a.P.max_events)) {
		a.ev_off[jb.out] = -1; a.n_ev[jb.out] = 0;
		return false;
	}
	const int r = jb.region;
	const long long rr0 = a.region_read_off[r], rr1 = a.region_read_off[r + 1];
	const long long ge0 = rr0 + 64 < rr1 ? rr0 + 64 : rr1;
	R.q_off = jb.q_off; R.t_off = jb.t_off; R.rr0 = rr0; R.rr1 = rr1;
	R.base0 = a.read_off[rr0]; R.end0 = a.read_off[ge0];
	R.coff = coff; R.qlen = jb.qlen; R.tlen = jb.tlen; R.region = r; R.out = jb.out; R.n_cigar = ez.n_cigar; R.job = j;
	R.ctg_rel = (int)(a.ctg_start[jb.out] - a.ref_origin[r]);
	R.nev = nev; R.ntrunc = ntrunc;
	int packed = a.pk != nullptr;
	if (packed) {
		// (eight flags a load, four loads in flight: a byte at a time this loop was a hundred dependent-looking trips per thread)
		typedef unsigned long long u64_unaligned_t __attribute__((aligned(1)));
		unsigned long long bad = 0;
		long long i = rr0;
		for (; i + 32 <= rr1; i += 32) {
			const unsigned long long b0 = *(const u64_unaligned_t *)(a.read_bad + i), b1 = *(const u64_unaligned_t *)(a.read_bad + i + 8);
			const unsigned long long b2 = *(const u64_unaligned_t *)(a.read_bad + i + 16), b3 = *(const u64_unaligned_t *)(a.read_bad + i + 24);
			bad |= (b0 | b1) | (b2 | b3);
		}
		for (; i + 8 <= rr1; i += 8) bad |= *(const u64_unaligned_t *)(a.read_bad + i);
		for (; i < rr1; ++i) bad |= a.read_bad[i];
		packed = bad == 0;
	}
	R.packed = packed;
	return true;
}

__global__ __launch_bounds__(256) void k_tally_prep(const TallyArgs a)
{
	const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (j >= *a.n_jobs || j >= a.njobs_cap) return;
	TallyRec R;
#pragma unroll
	for (int i = 0; i < 8; ++i) R.cig[i] = 0;
	if (!tally_make_rec(a, j, R)) return;
	const int idx = atomicAdd(&a.n_recs[0], 1);
	if (idx < a.rec_cap) {
		uint4 *dst = (uint4 *)&a.recs[idx];
		const uint4 *src = (const uint4 *)&R;
#pragma unroll
		for (int i = 0; i < 8; ++i) dst[i] = src[i];
	} else a.ovf_jobs[atomicAdd(&a.n_recs[1], 1)] = j;
}

template <int MINW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MINW, 8))) void k_tally(const TallyArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t tally_lds[];
	const int lane = lane_id();
	mark_start(a.t_start);
	const int nrec_all = uni(a.n_recs[0]), nrec = nrec_all < a.rec_cap ? nrec_all : a.rec_cap, njobs = nrec + uni(a.n_recs[1]);
	// Items (the jobs with events, k_tally_prep's records) are dealt round robin (wave w takes w, w + grid, ...): they are short
	// and alike, and a shared queue costs every wave a sweep over its eight counters at the end -- 65 000 same-address requests
	// per launch, which at the ~40 per microsecond one address sustains took longer than the items themselves.
	for (int it = (int)blockIdx.x; it < njobs; it += (int)gridDim.x) {
		const long long tj0 = a.prof ? (long long)clock64() : 0;
		// the record: one line, a dword per lane (a job that found the array full: the same header, worked out here)
		long long q_off, t_off, rr0, rr1, base0, end0, coff;
		int qlen, tlen, r, out, n_cigar, j, ctg_rel, nev, ntrunc; bool packed;
		CigSrc cig;
		if (it < nrec) {
			const unsigned *rw = (const unsigned *)&a.recs[it];
			const unsigned w = lane < 24 ? rw[lane] : 0u, cw8 = lane < 8 ? rw[24 + lane] : 0u;
			auto f32 = [&](int k) { return __builtin_amdgcn_readlane((int)w, k); };
			auto f64 = [&](int k) { return (long long)(((unsigned long long)(unsigned)f32(k + 1) << 32) | (unsigned)f32(k)); };
			q_off = f64(0); t_off = f64(2); rr0 = f64(4); rr1 = f64(6); base0 = f64(8); end0 = f64(10); coff = f64(12);
			qlen = f32(14); tlen = f32(15); r = f32(16); out = f32(17); n_cigar = f32(18); j = f32(19); ctg_rel = f32(20);
			nev = f32(21); ntrunc = f32(22); packed = f32(23) != 0;
			cig.mem = a.cig_pool + coff;
			cig.in_lanes = n_cigar <= 64;
			cig.cw = n_cigar <= 8 ? cw8 : (cig.in_lanes && lane < n_cigar) ? cig.mem[lane] : 0u;
		} else {
			TallyRec R;
#pragma unroll
			for (int i = 0; i < 8; ++i) R.cig[i] = 0;
			(void)tally_make_rec(a, uni(a.ovf_jobs[it - nrec]), R);   // (it has events: k_tally_prep said so)
			q_off = uni(R.q_off); t_off = uni(R.t_off); rr0 = uni(R.rr0); rr1 = uni(R.rr1); base0 = uni(R.base0); end0 = uni(R.end0); coff = uni(R.coff);
			qlen = uni(R.qlen); tlen = uni(R.tlen); r = uni(R.region); out = uni(R.out); n_cigar = uni(R.n_cigar); j = uni(R.job); ctg_rel = uni(R.ctg_rel);
			nev = uni(R.nev); ntrunc = uni(R.ntrunc); packed = uni(R.packed) != 0;
			cig.mem = a.cig_pool + coff;
			cig.in_lanes = n_cigar <= 64;
			cig.cw = (cig.in_lanes && lane < n_cigar) ? cig.mem[lane] : 0u;
		}
		long long eoff = (long long)j * a.P.max_events;         // the job's own slots: no atomic
		const long long tj1 = a.prof ? (long long)clock64() : 0;
		if (a.prof && lane == 0) atomicAdd((unsigned long long *)&a.prof[20], (unsigned long long)(tj1 - tj0));
		if (eoff + nev <= a.ev_pool_cap) {
			fill_events(cig, ntrunc, a.out_seq + q_off, qlen,
			            ctg_rel, a.ref_bases + t_off, tlen,
			            a.bases, a.read_off, a.mapq, rr0, rr1,
			            a.P, a.ev_pool + eoff, tally_lds, a.lds_bytes, j, (int)eoff, a.fb_items, a.fb_count,
			            a.hit_pool, a.hit_cursor, a.hit_cap, a.hit_overflow, a.hit_region_cnt ? a.hit_region_cnt + r : nullptr,
			            8 * rr0, a.hit_bump0, base0, end0, packed ? a.pk : nullptr);
		} else {
			if (lane == 0) atomicExch(&a.overflow[2], 1);
			eoff = -1; nev = 0;
		}
		if (lane == 0) { a.ev_off[out] = eoff; a.n_ev[out] = nev; }
		WSYNC();
		if (a.prof && lane == 0) {
			atomicAdd((unsigned long long *)&a.prof[nev ? 16 : 17], (unsigned long long)((long long)clock64() - tj0));
			atomicAdd((unsigned long long *)&a.prof[nev ? 18 : 19], 1ull);
		}
	}
}

// --------------------------------------------------------- alignment fallback
// indelope.nim:312-372.  For an event whose k-mer tally found both k-mers in one read, every read of the region
// (mapq >= 10, quality-trimmed) is aligned -- gap open 5, unbanded, no z-drop (ksw2.nim:159 defaults) -- to the
// reference window and to the contig, both cut at the read's start; count_flanked_cigar (:185-199) of the two
// truncated CIGARs decides the vote.  One wave per (event, read): work item j = read * n_events + event.
struct FbArgs {
	const FbItem *items; const int *n_items; int max_region_reads;
	const AlnJob *jobs;
	const uint8_t *out_seq, *ref_bases, *bases, *quals, *mapq;
	const int *trim_lo, *trim_hi;
	const long long *read_off, *region_read_off, *read_start, *ref_origin, *ctg_start;
	DevEvent *ev_pool;
	KswParams P; int min_mapq, trim_min_qual;
	int lds_budget;
	uint8_t *p_scratch; size_t p_cap;          // per workgroup
	uint32_t *cig_tmp; int cig_cap;            // per workgroup
	int *overflow;                             // [1] LDS / traceback scratch budget
	int *work_counter;
	unsigned long long *t_start;               // optional: see mark_start()
	int duo;                                   // 1: items that fit it take the two-target sweep (ksw_duo.h)
	// An item whose alignments need more LDS or traceback scratch than this launch gives a wave -- a read of an event on a contig far
	// longer than its reference window -- is put on ovf_list (when there is one); a second launch of this kernel -- in_list = that
	// list, a few workgroups with all the LDS and a large scratch each -- takes it.  Without ovf_list, or past ovf_cap, overflow[1].
	const int *in_list; const int *n_in;       // work item -> item index (null: the items themselves)
	int *ovf_list, *ovf_n; int ovf_cap;
};

// count_flanked_cigar (indelope.nim:185-199) over Ez.cigar (ksw2.nim:22-33); wave-uniform
__device__ __forceinline__ int count_flanked_cigar_dev(const uint32_t *cigar, int n_cigar, int max_q)
{
	const uint32_t max_off = (uint32_t)max_q;
	uint32_t off = 0;
	int matched = 0, n = 0, last_op = 0;
	for (int i = 0; i < n_cigar; ++i) {
		if (off >= max_off) break;
		const uint32_t c = (uint32_t)uni((int)cigar[i]), op = c & 0xf, len = c >> 4;
		if (op != 2) off += len;
		if (!matched) { if (op == 0) { n += 1; matched = 1; } }
		else n += 1;
		last_op = (int)op;
	}
	if (last_op != 0) n -= 1;
	return n;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_fallback(const FbArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	__shared__ int s_item;
	const int lane = lane_id();
	mark_start(a.t_start);
	const int n_items = *a.n_items;
	const long long total = (long long)n_items * a.max_region_reads;
	int n = total > 0x7fffffff ? 0x7fffffff : (int)total;
	if (a.in_list) { n = *a.n_in; n = n < a.ovf_cap ? n : a.ovf_cap; }
	uint8_t *p = a.p_scratch + (size_t)blockIdx.x * a.p_cap;
	uint32_t *ct = a.cig_tmp + (size_t)blockIdx.x * a.cig_cap;
	unsigned wq_dead = 0;
	for (;;) {
		if (lane == 0) s_item = wq_next(a.work_counter, n, (int)blockIdx.x, wq_dead);
		WSYNC();
		int j = uni(s_item);
		WSYNC();
		if (j < 0) break;
		if (a.in_list) j = uni(a.in_list[j]);
		auto punt = [&]() {                                              // not in this launch: the roomy one, or IHP_E_CAPACITY
			if (lane != 0) return;
			const int k = a.ovf_list ? atomicAdd(a.ovf_n, 1) : a.ovf_cap;
			if (k < a.ovf_cap) a.ovf_list[k] = j; else atomicExch(&a.overflow[1], 1);
		};
		// item j = read * n_events + event: the queue deals items to its shards by j % 64, and "read i of every event"
		// in one shard would put all the reads that start right of their event (skipped at once) together
		const int i = j / n_items, f = j - i * n_items;
		const FbItem it = a.items[f];
		const AlnJob jb = a.jobs[it.job];
		const int r = uni(jb.region);
		const long long r0 = uni(a.region_read_off[r]), r1 = uni(a.region_read_off[r + 1]);
		const long long ri = r0 + i;
		if (ri >= r1) continue;
		if ((int)a.mapq[ri] < a.min_mapq) continue;                    // :325
		const long long off = uni(a.read_off[ri]);
		const int len = (int)(uni(a.read_off[ri + 1]) - off);
		int lo = 0, hi = len, ta = 0;
		if (a.trim_lo) {
			lo = uni(a.trim_lo[ri]); hi = uni(a.trim_hi[ri]);
			lo = lo < 0 ? 0 : lo > len ? len : lo; hi = hi > len ? len : hi; hi = hi < lo ? lo : hi;
			ta = lo;
		} else if (a.quals) ta = read_trim_dev(a.quals + off, len, a.trim_min_qual, lo, hi);   // :328
		else if (len == 1) hi = 0;                                     // no qualities = all 255 (:28-30)
		const long long origin = uni(a.ref_origin[r]);
		DevEvent *E = a.ev_pool + it.ev;
		const int tstart = uni(E->tstart_rel), tstop = uni(E->tstop_rel);
		const long long rs = uni(a.read_start[ri]) + ta - origin;      // relative to the region origin, like tloc
		const int rl = hi - lo;
		if (rs > tstop) continue;                                      // :329
		const int L = uni((int)E->type) == 0 ? uni((int)E->len) : 0;   // :330-332
		if (rs + rl + L < tstart) continue;                            // :333
		const long long ctg_rel = uni(a.ctg_start[jb.out]) - origin;
		const long long start = (rs > ctg_rel ? rs : ctg_rel) - ctg_rel;   // :336
		const int ctg_len = uni(jb.qlen), reflen = uni(jb.tlen);
		const int rsub = start < reflen ? (int)(reflen - start) : 0;   // :337
		const int csub = start < ctg_len ? (int)(ctg_len - start) : 0; // :338
		const int tmax = rsub > csub ? rsub : csub;
		int w = a.P.w;
		if (w < 0) w = tmax > rl ? tmax : rl;
		int ncol_ = rl < tmax ? rl : tmax;
		ncol_ = ((ncol_ < w + 1 ? ncol_ : w + 1) + 15) / 16 + 1;
		const size_t pneed = ((size_t)(rl + tmax - 1 > 0 ? rl + tmax - 1 : 0) * ncol_ + 1) * 16;
		if (rl > 0 && tmax > 0 && (pneed > a.p_cap || rl + tmax + 8 > a.cig_cap)) {
			punt();
			continue;
		}
		const uint8_t *qy = a.bases + off + lo;
		// read vs reference window (:340), then read vs contig (:341); the register-resident ring sweep (ksw_wide.h)
		// when the read fits it, the generic LDS sweep otherwise
		const bool wide_ok = !(a.P.flag & KSW_EZ_RIGHT);
		int cnt[2];
		bool over = false;
		// both alignments in one sweep (ksw_duo.h) when the item is the usual kind: a read of at most 320 bases, both strings non-empty
		bool duo = false;
		if ((a.duo & 1) && rl > 0 && rsub > 0 && csub > 0 && ksw_duo_ok(a.P, rl, rsub, csub) &&
		    ksw_duo_lds_bytes(tmax) <= (size_t)a.lds_budget && ksw_duo_p_bytes(rl, tmax) <= a.p_cap) {
			DuoResult R;
			const uint8_t *tr = a.ref_bases + uni(jb.t_off) + start, *tc = a.out_seq + uni(jb.q_off) + start;
			const bool skip = (a.duo & 2) != 0;                            // ihp_debug_set("fb_skip", 0): every slot on every diagonal, as in round 5
			duo = rl <= 192 ? ksw_duo_sweep<3>(qy, rl, tr, rsub, tc, csub, a.P, lds, p, R, skip) : ksw_duo_sweep<5>(qy, rl, tr, rsub, tc, csub, a.P, lds, p, R, skip);
			if (duo) {
				KswOut o;
				ksw_duo_cigar<0>(R, p, rl, rsub, a.P.w, a.P.flag, ct, a.cig_cap, o);
				WSYNC();
				cnt[0] = o.n_cigar > 0 ? count_flanked_cigar_dev(ct, o.n_cigar, o.max_q) : 0;
				WSYNC();
				ksw_duo_cigar<1>(R, p, rl, csub, a.P.w, a.P.flag, ct, a.cig_cap, o);
				WSYNC();
				cnt[1] = o.n_cigar > 0 ? count_flanked_cigar_dev(ct, o.n_cigar, o.max_q) : 0;
				WSYNC();
			}
		}
		for (int side = 0; side < 2 && !duo; ++side) {
			const int tl = side ? csub : rsub;
			const uint8_t *tg = side ? a.out_seq + uni(jb.q_off) + (csub ? start : 0) : a.ref_bases + uni(jb.t_off) + (rsub ? start : 0);
			KswOut o;
			o.n_cigar = 0; o.max_q = -1;
			if (rl > 0 && tl > 0) {
				if (wide_ok && ksw_wide_ok<3>(a.P, rl, tl) && ksw_wide_lds_bytes<3>(rl, tl) <= (size_t)a.lds_budget)
					ksw_wave_wide<3, false>(qy, rl, tg, tl, a.P, lds, p, ct, a.cig_cap, o);
				else if (wide_ok && ksw_wide_ok<6>(a.P, rl, tl) && ksw_wide_lds_bytes<6>(rl, tl) <= (size_t)a.lds_budget)
					ksw_wave_wide<6, false>(qy, rl, tg, tl, a.P, lds, p, ct, a.cig_cap, o);
				else if (ksw_lds_bytes(rl, tl) <= (size_t)a.lds_budget)
					ksw_wave(qy, rl, tg, tl, a.P, lds, p, ct, a.cig_cap, o);
				else over = true;
			}
			WSYNC();
			cnt[side] = o.n_cigar > 0 ? count_flanked_cigar_dev(ct, o.n_cigar, o.max_q) : 0;   // :343-344
			WSYNC();
		}
		if (over) { punt(); continue; }
		const int rn = cnt[0], an = cnt[1];
		if (lane == 0) {
			if (rn == 1 && an > 1) atomicAdd(&E->ref_support, 1);      // :353-354
			else if (an == 1 && rn > 1) atomicAdd(&E->alt_support, 1); // :355-356
		}
	}
}

// Diagnostics (ihp_debug_ksw_duo_batch): the two-target sweep of ksw_duo.h on caller-made (read, target 0, target 1) items, so
// that tests can hold it against the compiled reference directly.  One wavefront per item; an item the sweep does not take
// (ksw_duo_ok) gets n_cigar = -2 in both records.
struct DuoTestArgs {
	int n;
	const uint8_t *q, *t0, *t1; const long long *q_off, *t0_off, *t1_off;
	KswParams P; int lds_budget;
	uint8_t *p_scratch; size_t p_cap;          // per workgroup
	uint32_t *cig_tmp; int cig_cap;            // per workgroup
	KswOut *ez;                                // [2 n]
	uint32_t *cig; int cig_slot;               // [2 n][cig_slot]
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_ksw_duo_test(const DuoTestArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const int lane = lane_id();
	uint8_t *p = a.p_scratch + (size_t)blockIdx.x * a.p_cap;
	uint32_t *ct = a.cig_tmp + (size_t)blockIdx.x * a.cig_cap;
	for (int i = (int)blockIdx.x; i < a.n; i += (int)gridDim.x) {
		const int ql = (int)uni(a.q_off[i + 1] - a.q_off[i]), tl0 = (int)uni(a.t0_off[i + 1] - a.t0_off[i]), tl1 = (int)uni(a.t1_off[i + 1] - a.t1_off[i]);
		const int tmax = tl0 > tl1 ? tl0 : tl1;
		KswOut o[2];
		for (int k = 0; k < 2; ++k) {
			o[k].max = 0; o[k].zdropped = 0; o[k].max_q = o[k].max_t = o[k].mqe_t = o[k].mte_q = -1;
			o[k].mqe = o[k].mte = o[k].score = KSW_NEG_INF; o[k].n_cigar = -2;
		}
		bool done = false;
		if (ql > 0 && tl0 > 0 && tl1 > 0 && ksw_duo_ok(a.P, ql, tl0, tl1) && ksw_duo_lds_bytes(tmax) <= (size_t)a.lds_budget &&
		    ksw_duo_p_bytes(ql, tmax) <= a.p_cap && ql + tmax + 8 <= a.cig_cap && ql + tmax + 8 <= a.cig_slot) {
			DuoResult R;
			const uint8_t *qy = a.q + uni(a.q_off[i]), *tg0 = a.t0 + uni(a.t0_off[i]), *tg1 = a.t1 + uni(a.t1_off[i]);
			done = ql <= 192 ? ksw_duo_sweep<3>(qy, ql, tg0, tl0, tg1, tl1, a.P, lds, p, R) : ksw_duo_sweep<5>(qy, ql, tg0, tl0, tg1, tl1, a.P, lds, p, R);
			if (done) {
				ksw_duo_cigar<0>(R, p, ql, tl0, a.P.w, a.P.flag, ct, a.cig_cap, o[0]);
				WSYNC();
				for (int j = lane; j < o[0].n_cigar; j += 64) a.cig[(size_t)(2 * i) * a.cig_slot + j] = ct[j];
				WSYNC();
				ksw_duo_cigar<1>(R, p, ql, tl1, a.P.w, a.P.flag, ct, a.cig_cap, o[1]);
				WSYNC();
				for (int j = lane; j < o[1].n_cigar; j += 64) a.cig[(size_t)(2 * i + 1) * a.cig_slot + j] = ct[j];
				WSYNC();
			}
		}
		if (lane == 0) { a.ez[2 * i] = o[0]; a.ez[2 * i + 1] = o[1]; }
		WSYNC();
	}
}

// ------------------------------------------------------------ the compact slab (ihp_slab2_layout) -> ihp_batch_in's arrays
// One wavefront per region: read_off as the prefix sum of the lengths behind region_base_off[r], read_start / read_stop from
// the 32-bit start relative to the window's origin and the 16-bit span, the trim bounds and the two bytes widened, the window's
// bases from 4 (BAM's code) or 2 bits each to ASCII.  A region whose lengths do not add up to its region_base_off step raises
// *bad (reported as IHP_E_ARG by the first wait for the batch).
struct SlabExpandArgs {
	int n_regions; long long n_reads;
	const long long *region_read_off, *region_base_off, *ref_off, *ref_origin;
	const int *start_rel; const unsigned short *len, *span, *trim_lo_in, *trim_hi_in; const uint8_t *mapq_in, *rflags;
	const uint8_t *ref_packed; int ref_2bit;
	long long *read_off, *read_start, *read_stop; int *trim_lo, *trim_hi; uint8_t *mapq, *read_skip, *ref_bases;
	int *bad;
};

__global__ __launch_bounds__(64) void k_slab_expand(const SlabExpandArgs a)
{
	const int lane = lane_id();
	for (int r = (int)blockIdx.x; r < a.n_regions; r += (int)gridDim.x) {
		const long long r0 = uni(a.region_read_off[r]), r1 = uni(a.region_read_off[r + 1]);
		long long off = uni(a.region_base_off[r]);
		const long long off_end = uni(a.region_base_off[r + 1]);   // (no read_off leaves the region's bases, whatever the lengths say)
		const long long origin = uni(a.ref_origin[r]);
		for (long long g0 = r0; g0 < r1; g0 += 64) {
			const long long i = g0 + lane;
			const bool live = i < r1;
			const unsigned ln = live ? a.len[i] : 0u;
			const unsigned incl = wave_scan_add(ln);
			if (live) {
				{ const long long ro = off + (long long)(incl - ln); a.read_off[i] = ro < off_end ? ro : off_end; }
				const long long st = origin + (long long)a.start_rel[i];
				a.read_start[i] = st; a.read_stop[i] = st + (long long)a.span[i];
				a.trim_lo[i] = (int)a.trim_lo_in[i]; a.trim_hi[i] = (int)a.trim_hi_in[i];
				a.mapq[i] = a.mapq_in[i]; a.read_skip[i] = a.rflags[i] & 1;
			}
			off += (long long)(unsigned)__builtin_amdgcn_readlane((int)incl, 63);
		}
		if (off != off_end && lane == 0) atomicExch(a.bad, 1);
		if (r == a.n_regions - 1 && lane == 0) a.read_off[a.n_reads] = off_end;
		// the window
		const long long f0 = uni(a.ref_off[r]), f1 = uni(a.ref_off[r + 1]);
		if (a.ref_2bit) {
			const uint8_t *src = a.ref_packed + (f0 >> 2) + r;
			for (long long j = lane; j < f1 - f0; j += 64) a.ref_bases[f0 + j] = (uint8_t)"ACGT"[(src[j >> 2] >> (2 * (j & 3))) & 3];
		} else {
			const uint8_t *src = a.ref_packed + (f0 >> 1) + r;
			for (long long j = lane; j < f1 - f0; j += 64) a.ref_bases[f0 + j] = (uint8_t)"=ACMGRSVTWYHKDBN"[(src[j >> 1] >> ((j & 1) ? 0 : 4)) & 15];
		}
	}
	if (a.n_regions == 0 && blockIdx.x == 0 && lane == 0) a.read_off[0] = 0;
}

// ------------------------------------------------------------------- summary
struct SummaryArgs {
	int n_regions;
	const long long *region_read_off;
	const int *status, *n_pre, *n_final, *aln_flags, *n_ev;
	const long long *ev_off;
	const DevEvent *ev_pool;
	ihp_region_summary *out;
	// end-of-run housekeeping (this is the last kernel of a run's launch chain): the run's counters, overflow flags
	// and stamps are copied to `report` (page-locked host memory: ihp_batch_sync reads them without a copy) and
	// everything the next run expects to be zero is cleared here, so a run needs no memset
	int *zero; int n_zero;           // counters | stamps | work queues | per-region hit counts, in ints
	int *report; int n_report;       // the first n_report ints of `zero`
	int sticky;                      // the one int of `zero` that is reported and NOT cleared: raised once per batch (k_slab_expand), it has to
	                                 // be in every run's report until a wait has latched it (upload, run, run, sync; a run cut short)
	int t_end;                       // >= 0: the two ints at this index of `report` get the device wall clock at the start of this
	                                 // launch (= the end of the stage before it, in stream order)
};

__global__ void k_summary(const SummaryArgs a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long t_now = a.t_end >= 0 ? (unsigned long long)wall_clock64() : 0ull;
	if (a.zero) {
		const int nt = (int)(gridDim.x * blockDim.x);
		for (int i = r; i < a.n_zero; i += nt) {
			if (i < a.n_report) a.report[i] = a.zero[i];
			if (i != a.sticky) a.zero[i] = 0;
		}
		if (a.t_end >= 0 && r == a.t_end) { a.report[r] = (int)(unsigned)t_now; a.report[r + 1] = (int)(unsigned)(t_now >> 32); }
	}
	if (r >= a.n_regions) return;
	ihp_region_summary s;
	s.status = a.status[r]; s.n_contigs_pre = a.n_pre[r]; s.n_contigs = a.n_final[r];
	s.n_aligned = 0; s.n_events = 0; s.n_tallied = 0; s.ref_support = -1; s.alt_support = -1;
	const long long base = a.region_read_off[r];
	for (int k = 0; k < s.n_contigs; ++k) {
		const long long slot = base + k;
		if (!(a.aln_flags[slot] & IHP_ALN_DONE)) continue;
		s.n_aligned++;
		const int ne = a.n_ev[slot];
		s.n_events += ne;
		for (int e = 0; e < ne; ++e) {
			const DevEvent &E = a.ev_pool[a.ev_off[slot] + e];
			if (E.status == IHP_EV_TALLIED) {
				if (s.n_tallied == 0) { s.ref_support = E.ref_support; s.alt_support = E.alt_support; }
				s.n_tallied++;
			}
		}
	}
	a.out[r] = s;
}

// ---------------------------------------------------------------------- pack
// Results are produced slot-indexed (contig k of region r at region_read_off[r] + k, bases at the region's read
// offset, ...), so that no kernel needs a data-dependent size.  Fetching that layout would move the read-sized
// arrays over PCIe; instead three small kernels compact it on the device into the flat `ihp_batch_out` arrays,
// laid out in one slab that is copied to the host in one piece.
//   k_pack_count: per region -> contigs, bases, CIGAR words, events;  k_pack_scan: exclusive prefix sums;
//   k_pack: one wave per region copies its contigs to their final places.
struct PackCountArgs {
	int R;
	const long long *region_read_off;
	const int *n_final, *ctg_len, *aln_flags, *n_ev;
	const KswOut *ez;
	const long long *ev_off; const DevEvent *ev_pool;
	long long *cnt;                 // [5][R+1]: contigs, bases, cigar words, events, hit entries
};

__global__ void k_pack_count(const PackCountArgs a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.R) return;
	const long long base = a.region_read_off[r];
	const int n = a.n_final[r];
	const long long nr = a.region_read_off[r + 1] - base;
	long long B = 0, W = 0, E = 0, Hn = 0;
	for (int k = 0; k < n; ++k) {
		const long long sl = base + k;
		B += a.ctg_len[sl];
		if (a.aln_flags[sl] & IHP_ALN_DONE) {
			const int nc = a.ez[sl].n_cigar, ne = a.n_ev[sl];
			W += nc > 0 ? nc : 0; E += ne;
			for (int e = 0; e < ne; ++e) if (a.ev_pool[a.ev_off[sl] + e].hit_off >= 0) Hn += nr;
		}
	}
	const size_t S = (size_t)a.R + 1;
	a.cnt[r] = n; a.cnt[S + r] = B; a.cnt[2 * S + r] = W; a.cnt[3 * S + r] = E; a.cnt[4 * S + r] = Hn;
}

// in place: counts -> exclusive prefix sums, totals at index R.  One workgroup of 16 waves: every wave owns a contiguous
// sixteenth of the regions and walks it 64 at a time (coalesced; the first version gave every thread ten consecutive entries and
// had five threads add up 1024 partial sums one LDS round trip at a time: 68 us on the critical path of every fetched batch).
__device__ __forceinline__ long long wave_incl_scan_ll(long long v)
{
	const int lane = lane_id();
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const long long u = __shfl_up(v, d, 64);
		if (lane >= d) v += u;
	}
	return v;
}
__global__ __launch_bounds__(1024) void k_pack_scan(int R, long long *cnt, long long *host_tot)
{
	__shared__ long long part[5][16];
	const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
	const long long per = (((long long)R + 15) / 16 + 63) / 64 * 64;
	const long long lo = (long long)w * per < R ? (long long)w * per : R, hi = lo + per < R ? lo + per : R;
	const size_t S = (size_t)R + 1;
	for (int a = 0; a < 5; ++a) {
		long long s = 0;
		for (long long i = lo + lane; i < hi; i += 64) s += cnt[a * S + i];
		s = wave_incl_scan_ll(s);
		if (lane == 63) part[a][w] = s;
	}
	__syncthreads();
	if (t < 5) {
		long long run = 0;
		for (int i = 0; i < 16; ++i) { const long long c = part[t][i]; part[t][i] = run; run += c; }
		cnt[t * S + R] = run;
		if (host_tot) host_tot[t] = run;                     // page-locked host memory: the caller sizes the result slab from these
	}
	__syncthreads();
	for (int a = 0; a < 5; ++a) {
		long long run = part[a][w];
		for (long long i0 = lo; i0 < hi; i0 += 64) {
			const long long i = i0 + lane;
			const long long c = i < hi ? cnt[a * S + i] : 0;
			const long long incl = wave_incl_scan_ll(c);
			if (i < hi) cnt[a * S + i] = run + incl - c;
			run += __shfl(incl, 63, 64);
		}
	}
}

struct PackArgs {
	int R;
	const long long *region_read_off, *ref_origin;
	const int *status, *n_pre, *n_final, *ctg_len, *aln_flags, *aln_ref_len, *n_ev;
	const long long *ctg_start, *ctg_nreads, *ctg_seq_off, *aln_ref_start, *cig_off, *ev_off;
	const uint8_t *out_seq; const uint32_t *out_sup; const KswOut *ez; const uint32_t *cig_pool; const DevEvent *ev_pool;
	const int *hit_pool;
	const long long *cnt;           // prefix sums of k_pack_scan
	// the flat arrays of ihp_batch_out (device slab)
	int32_t *o_status, *o_n_pre; int64_t *o_contig_off;
	int64_t *o_ctg_start, *o_ctg_nreads, *o_ctg_seq_off; uint8_t *o_seq; uint32_t *o_sup;
	int32_t *o_aln_flags; int64_t *o_aln_ref_start; int32_t *o_aln_ref_len; ihp_ez *o_ez;
	int64_t *o_cigar_off; uint32_t *o_cigar; int64_t *o_event_off; ihp_event *o_events;
	int64_t *o_hit_off; int32_t *o_ref_hit, *o_alt_hit;
	// IHP_FETCH_COMPACT: bases 4 bits each (contig c from byte (bb >> 1) + c), supports a byte each, 255 = see the escapes
	// (esc_idx[0]: their number, counted here; esc_val[0]: raised when a base has no 4-bit code; entries from 1)
	uint8_t *o_seq4, *o_sup8; long long *esc_idx; unsigned *esc_val; long long esc_cap;
};
// ASCII -> BAM's 4-bit code ("=ACMGRSVTWYHKDBN"), 255: none
__device__ __forceinline__ unsigned code4_of(unsigned ch)
{
	unsigned c;
	switch (ch) {
	case '=': c = 0; break; case 'A': c = 1; break; case 'C': c = 2; break; case 'M': c = 3; break;
	case 'G': c = 4; break; case 'R': c = 5; break; case 'S': c = 6; break; case 'V': c = 7; break;
	case 'T': c = 8; break; case 'W': c = 9; break; case 'Y': c = 10; break; case 'H': c = 11; break;
	case 'K': c = 12; break; case 'D': c = 13; break; case 'B': c = 14; break; case 'N': c = 15; break;
	default: c = 255;
	}
	return c;
}

__global__ __launch_bounds__(64) void k_pack(const PackArgs a)
{
	const int lane = lane_id();
	const size_t S = (size_t)a.R + 1;
	for (int r = (int)blockIdx.x; r <= a.R; r += (int)gridDim.x) {
		long long c = a.cnt[r], bb = a.cnt[S + r], wd = a.cnt[2 * S + r], ev = a.cnt[3 * S + r], hn = a.cnt[4 * S + r];
		if (r == a.R) {                                       // the closing entries of the offset arrays
			if (lane == 0) { a.o_contig_off[r] = c; a.o_ctg_seq_off[c] = bb; a.o_cigar_off[c] = wd; a.o_event_off[c] = ev; a.o_hit_off[ev] = hn; }
			break;
		}
		const long long base = a.region_read_off[r], origin = a.ref_origin[r];
		const long long nr = a.region_read_off[r + 1] - base;
		const int n = a.n_final[r];
		if (lane == 0) { a.o_status[r] = a.status[r]; a.o_n_pre[r] = a.n_pre[r]; a.o_contig_off[r] = c; }
		for (int k = 0; k < n; ++k, ++c) {
			const long long sl = base + k;
			const int len = a.ctg_len[sl], flags = a.aln_flags[sl];
			const bool done = (flags & IHP_ALN_DONE) != 0;
			const long long so = a.ctg_seq_off[sl];
			if (a.o_seq) for (int i = lane; i < len; i += 64) { a.o_seq[bb + i] = a.out_seq[so + i]; a.o_sup[bb + i] = a.out_sup[so + i]; }
			if (a.o_seq4) {
				uint8_t *d4 = a.o_seq4 + (bb >> 1) + c;
				bool bad = false;
				for (int j = lane; 2 * j < len; j += 64) {
					const unsigned c0 = code4_of(a.out_seq[so + 2 * j]), c1 = 2 * j + 1 < len ? code4_of(a.out_seq[so + 2 * j + 1]) : 0u;
					bad |= c0 > 15u || c1 > 15u;
					d4[j] = (uint8_t)((c0 & 15u) << 4 | (c1 & 15u));
				}
				if (ballot(bad) && lane == 0) atomicExch(a.esc_val, 1u);
				for (int i = lane; i < len; i += 64) {
					const uint32_t s = a.out_sup[so + i];
					a.o_sup8[bb + i] = (uint8_t)(s < 255u ? s : 255u);
					if (s >= 255u) {
						const long long k = (long long)atomicAdd((unsigned long long *)a.esc_idx, 1ull);
						if (k < a.esc_cap) { a.esc_idx[1 + k] = bb + i; a.esc_val[1 + k] = s; }
					}
				}
			}
			KswOut z;
			z.max = z.zdropped = z.max_q = z.max_t = z.mqe = z.mqe_t = z.mte = z.mte_q = z.score = z.n_cigar = 0;
			if (done) z = a.ez[sl];
			if (lane == 0) {
				a.o_ctg_start[c] = a.ctg_start[sl]; a.o_ctg_nreads[c] = a.ctg_nreads[sl]; a.o_ctg_seq_off[c] = bb;
				a.o_aln_flags[c] = flags; a.o_aln_ref_start[c] = a.aln_ref_start[sl]; a.o_aln_ref_len[c] = a.aln_ref_len[sl];
				a.o_cigar_off[c] = wd; a.o_event_off[c] = ev;
				ihp_ez o;
				o.max = z.max; o.zdropped = z.zdropped; o.max_q = z.max_q; o.max_t = z.max_t; o.mqe = z.mqe; o.mqe_t = z.mqe_t;
				o.mte = z.mte; o.mte_q = z.mte_q; o.score = z.score; o.n_cigar = z.n_cigar;
				a.o_ez[c] = o;
			}
			bb += len;
			if (!done) continue;
			if (z.n_cigar > 0) {
				const long long co = a.cig_off[sl];
				for (int i = lane; i < z.n_cigar; i += 64) a.o_cigar[wd + i] = a.cig_pool[co + i];
				wd += z.n_cigar;
			}
			const int ne = a.n_ev[sl];
			for (int en = lane; en < ne; en += 64) {
				const DevEvent d = a.ev_pool[a.ev_off[sl] + en];
				ihp_event x;
				x.tstart = origin + d.tstart_rel; x.tstop = origin + d.tstop_rel; x.qstart = d.qstart; x.qstop = d.qstop;
				x.len = d.len; x.type = d.type; x.status = d.status; x.fallback_needed = d.fallback; x.aligned = d.aligned;
				x.cf_offset = d.cf_offset; x.ref_support = d.ref_support; x.alt_support = d.alt_support; x.both_found = d.both_found;
				for (int i = 0; i < 32; ++i) { x.ref_kmer[i] = d.ref_kmer[i]; x.alt_kmer[i] = d.alt_kmer[i]; }
				x.gt = IHP_GT_UNKNOWN; x.kmer_ref_support = d.kmer_ref; x.kmer_alt_support = d.kmer_alt; x.kmer_both_found = d.kmer_both;
				x.gl[0] = x.gl[1] = x.gl[2] = 0; x.qual = 0;         // genotype(): host, fp64 (genotyper.nim:36-47)
				a.o_events[ev + en] = x;
			}
			for (int e = 0; e < ne; ++e) {                     // first-hit positions, in event order
				const long long ho = a.ev_pool[a.ev_off[sl] + e].hit_off;
				if (lane == 0) a.o_hit_off[ev + e] = hn;
				if (ho < 0) continue;
				for (long long i = lane; i < nr; i += 64) { a.o_ref_hit[hn + i] = a.hit_pool[ho + i]; a.o_alt_hit[hn + i] = a.hit_pool[ho + nr + i]; }
				hn += nr;
			}
			ev += ne;
		}
	}
}

// ------------------------------------------------ single-op kernels (Contig API)
struct OpArgs {
	int op;                      // 0 slide_align, 1 insert, 2 trim
	uint8_t *arena_seq; uint32_t *arena_sup; int arena_cap;
	Corr *corr; int corr_cap;
	int t_off, t_len, t_cap; long long t_nreads, t_start;
	int q_off, q_len, q_cap; long long q_nreads, q_start;
	long long min_overlap, max_mismatch; int rule;
	int off, ncorr; long long min_support;
	long long *result;           // [16]
};

__global__ __launch_bounds__(64) void k_contig_op(const OpArgs a)
{
	typedef RegionStateT<64> ST;
	__shared__ ST S;
	const int lane = lane_id();
	Arena A; A.seq = a.arena_seq; A.sup = a.arena_sup; A.cap = a.arena_cap; A.stage_off = a.arena_cap;
	A.corr = a.corr; A.corr_cap = a.corr_cap; A.prof = nullptr;
	for (int i = lane; i <= ST::MAXC; i += 64) S.alive[i] = 0;
	if (lane == 0) {
		S.off[0] = a.t_off; S.len[0] = a.t_len; S.cap[0] = a.t_cap; S.nreads[0] = a.t_nreads; S.start[0] = a.t_start; S.alive[0] = 1;
		S.off[1] = a.q_off; S.len[1] = a.q_len; S.cap[1] = a.q_cap; S.nreads[1] = a.q_nreads; S.start[1] = a.q_start; S.alive[1] = 1;
		S.bump = align4(a.q_off + a.q_cap); S.err = 0;
		S.lo3[0] = S.lo3[1] = 0x3fffffff; S.hi3[0] = S.hi3[1] = 0;
	}
	WSYNC();
	long long rc = 0, found = 0, ma = 0, mm = 0, off = 0, nc = 0;
	if (a.op == 0) {
		Best b = {0, 0, 0, -1, -1, 0};
		slide_scan(S, A, 1, 0, 0, (int)a.min_overlap, (int)a.max_mismatch, a.rule, b);
		found = b.found; ma = b.found ? b.ma : a.min_overlap - 1; mm = b.found ? b.mm : a.max_mismatch + 1; off = b.off;
		if (b.found) {
			const int n = emit_corrections(S, A, 1, 0, b.off, a.rule);
			if (n < 0) { rc = IHP_E_CAPACITY; nc = 0; } else nc = n;
		}
	} else if (a.op == 1) {
		rc = insert_dev(S, A, 0, 1, a.off, a.ncorr);
	} else {
		trim_dev(S, A, 0, a.min_support);
	}
	WSYNC();
	if (lane == 0) {
		long long *R = a.result;
		R[0] = rc; R[1] = found; R[2] = ma; R[3] = mm; R[4] = off; R[5] = nc;
		R[6] = S.off[0]; R[7] = S.len[0]; R[8] = S.nreads[0]; R[9] = S.start[0];
		R[10] = S.off[1]; R[11] = S.len[1]; R[12] = S.nreads[1]; R[13] = S.start[1];
	}
}

struct TallyOneArgs {
	const uint8_t *bases; const long long *read_off; const uint8_t *mapq;
	int n_reads, min_mapq, K;
	unsigned long long refe, alte;
	int *counts;
};

__global__ __launch_bounds__(64) void k_tally_one(const TallyOneArgs a)
{
	int counts[3];
	tally_reads(a.bases, a.read_off, a.mapq, 0, a.n_reads, a.min_mapq, a.K, a.refe, a.alte, counts);
	if (lane_id() == 0) { a.counts[0] = counts[0]; a.counts[1] = counts[1]; a.counts[2] = counts[2]; }
}

}  // namespace ihp
