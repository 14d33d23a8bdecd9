// roi_dev.h -- the ROI evidence scan of the BAM sweep: event_locations, gen_roi_internal
// and gen_roi of src/indelope.nim:430-445, :461-545, for one run of reads of one target at a time.
//
// The reference walks the reads once, keeps a cache of the reads since the last coverage gap and scans the
// evidence window of the cache when the next gap comes.  Nothing in that depends on the order of evaluation
// except where the windows are cut, so the device version works on the whole run at once:
//   * the cache is flushed at read i exactly when start[i] exceeds the stop of every earlier non-skippable
//     read (the cache's own maximum equals the global one: everything before the cache ended before it began),
//     so an exclusive prefix maximum of the stops marks the cuts (k_roi_pmax*, a three-phase scan);
//   * evidence[p] = min(255, number of non-match CIGAR ops of non-skippable reads that span p): one thread per
//     read, one 32-bit atomic add per covered position (k_roi_evidence);
//   * regions = maximal runs of evidence >= min_event_support, cut at the marked read starts (a window never
//     continues past the read that flushed it): starts and ends are found per position and compacted in
//     position order with a block-count scan (k_roi_count, k_roi_emit);
//   * the reads of a region are the first max_reads + 1 non-skippable reads, in BAM order, with
//     start <= roi_end and stop >= roi_start; the candidates lie between two binary searches (inclusive prefix
//     maximum of the stops >= roi_start from below, start > roi_end from above) and one wave per region walks
//     them 64 at a time, keeping order with a ballot (k_roi_reads).
// All of it is byte/integer traffic over the reads and the evidence array: HBM bound.
#pragma once
#include "ihp_common.h"

namespace ihp {

constexpr int ROI_BLOCK = 1024;
constexpr long long ROI_NONE = -0x7fffffffffffffffll - 1;

struct RoiArgs {
	long long n_reads, len;                               // len = span + 1 evidence entries
	const long long *start, *stop;                        // relative to origin
	const uint8_t *skip;
	const long long *cigar_off; const uint32_t *cigar;
	long long *pmax_incl;                                 // [n_reads] inclusive prefix max of the non-skippable stops
	long long *block_max;                                 // [blocks]
	unsigned *evidence;                                   // [len]
	uint8_t *cut;                                         // [len + 1]: a window starts here
	int min_evidence, min_reads, max_reads;
	// run compaction
	long long *block_cnt;                                 // [2][pos_blocks + 1]: starts, ends
	long long pos_blocks;
	long long *roi_start, *roi_end;                       // [n_roi]
	// reads of the regions
	long long n_roi;
	int *roi_cnt;                                         // [n_roi] overlapping reads, capped at max_reads + 1
	const long long *roi_off;                             // [n_roi] output offset, -1: region not yielded
	long long *roi_reads;
};

__device__ __forceinline__ long long ll_max(long long a, long long b) { return a > b ? a : b; }

// phase 1: per-block maximum of the non-skippable stops
__global__ __launch_bounds__(ROI_BLOCK) void k_roi_pmax_blocks(const RoiArgs a)
{
	__shared__ long long red[ROI_BLOCK / 64];
	const long long i = (long long)blockIdx.x * ROI_BLOCK + threadIdx.x;
	long long v = ROI_NONE;
	if (i < a.n_reads && !(a.skip && a.skip[i])) v = a.stop[i];
	for (int d = 32; d; d >>= 1) v = ll_max(v, __shfl_xor(v, d));
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		long long m = red[0];
		for (int k = 1; k < ROI_BLOCK / 64; ++k) m = ll_max(m, red[k]);
		a.block_max[blockIdx.x] = m;
	}
}

// phase 2: exclusive scan (max) of the block maxima, in place; one workgroup
__global__ __launch_bounds__(ROI_BLOCK) void k_roi_pmax_scan(long long *block_max, long long n_blocks)
{
	__shared__ long long part[ROI_BLOCK];
	const int t = (int)threadIdx.x;
	const long long per = (n_blocks + ROI_BLOCK - 1) / ROI_BLOCK;
	const long long lo = (long long)t * per < n_blocks ? (long long)t * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
	long long m = ROI_NONE;
	for (long long i = lo; i < hi; ++i) m = ll_max(m, block_max[i]);
	part[t] = m;
	__syncthreads();
	if (t == 0) {
		long long run = ROI_NONE;
		for (int i = 0; i < ROI_BLOCK; ++i) { const long long c = part[i]; part[i] = run; run = ll_max(run, c); }
	}
	__syncthreads();
	long long run = part[t];
	for (long long i = lo; i < hi; ++i) { const long long c = block_max[i]; block_max[i] = run; run = ll_max(run, c); }
}

// phase 3: inclusive prefix maximum per read, and the cuts: read i flushes the cache when its start lies beyond
// every earlier non-skippable stop (indelope.nim:529); the evidence window that follows begins at its start (:534)
__global__ __launch_bounds__(ROI_BLOCK) void k_roi_pmax_apply(const RoiArgs a)
{
	__shared__ long long sc[ROI_BLOCK];
	const int t = (int)threadIdx.x;
	const long long i = (long long)blockIdx.x * ROI_BLOCK + t;
	long long v = ROI_NONE;
	if (i < a.n_reads && !(a.skip && a.skip[i])) v = a.stop[i];
	sc[t] = v;
	__syncthreads();
	for (int d = 1; d < ROI_BLOCK; d <<= 1) {                // Hillis-Steele inclusive scan
		const long long o = t >= d ? sc[t - d] : ROI_NONE;
		__syncthreads();
		sc[t] = ll_max(sc[t], o);
		__syncthreads();
	}
	if (i >= a.n_reads) return;
	const long long before = a.block_max[blockIdx.x];
	const long long excl = ll_max(before, t ? sc[t - 1] : ROI_NONE);
	a.pmax_incl[i] = ll_max(before, sc[t]);
	const long long s = a.start[i];
	// (:529 also asks for a non-empty cache: behind a skippable read that flushed it, the next reads -- skippable or not --
	// flush nothing until a non-skippable one has been added, and the window keeps starting at that first read's start.
	// In a run of skippable reads the stops seen so far do not change and the starts ascend, so it is enough to look at the
	// read before.  Only min_event_support = 0 can tell: otherwise no evidence lies between the two starts.)
	bool first = true;
	if (i > 0 && a.skip && a.skip[i - 1]) first = !(a.start[i - 1] > excl);
	if (excl != ROI_NONE && s > excl && first && s >= 0 && s <= a.len) a.cut[s] = 1;
}

// event_locations (:430-445) of one read per thread, evidence[i] += 1 over every event (:539-543)
__global__ void k_roi_evidence(const RoiArgs a)
{
	const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.n_reads || (a.skip && a.skip[r])) return;
	const long long rs = a.start[r];
	long long off = 0;
	for (long long c = a.cigar_off[r]; c < a.cigar_off[r + 1]; ++c) {
		const unsigned w = a.cigar[c], op = w & 0xf;
		const long long n = (long long)(w >> 4);
		const bool cons = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;    // consumes.reference
		if (op != 0) {
			const long long es = rs + off, ee = cons ? es + n : es + 1;
			for (long long p = es < 0 ? 0 : es; p < ee && p < a.len; ++p) atomicAdd(&a.evidence[p], 1u);
		}
		if (cons) off += n;
	}
}

__device__ __forceinline__ bool roi_hot(const RoiArgs &a, long long p)
{   // evidence[p] >= min_evidence with the uint8 saturation of :541-543
	if (p < 0 || p >= a.len) return false;
	const unsigned e = a.evidence[p];
	return (e > 255u ? 255u : e) >= (unsigned)a.min_evidence;
}

// a region starts at p: hot, and the previous position is cold or p opens a new window; it ends at p: hot, and the
// next position is cold, or opens a new window, or is the end of the array
__device__ __forceinline__ void roi_edges(const RoiArgs &a, long long p, bool &is_start, bool &is_end)
{
	const bool hot = roi_hot(a, p);
	is_start = hot && (p == 0 || a.cut[p] || !roi_hot(a, p - 1));
	is_end = hot && (p == a.len - 1 || a.cut[p + 1] || !roi_hot(a, p + 1));
}

__global__ __launch_bounds__(ROI_BLOCK) void k_roi_count(const RoiArgs a)
{
	__shared__ int red[2][ROI_BLOCK / 64];
	const long long p = (long long)blockIdx.x * ROI_BLOCK + threadIdx.x;
	bool s = false, e = false;
	if (p < a.len) roi_edges(a, p, s, e);
	const int ns = popc64(ballot(s)), ne = popc64(ballot(e));
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ns; red[1][threadIdx.x >> 6] = ne; }
	__syncthreads();
	if (threadIdx.x == 0) {
		long long cs = 0, ce = 0;
		for (int k = 0; k < ROI_BLOCK / 64; ++k) { cs += red[0][k]; ce += red[1][k]; }
		a.block_cnt[blockIdx.x] = cs; a.block_cnt[a.pos_blocks + 1 + blockIdx.x] = ce;
	}
}

// exclusive sum scan of the two count arrays ([n + 1] each, total at index n); one workgroup
__global__ __launch_bounds__(ROI_BLOCK) void k_roi_count_scan(long long *cnt, long long n)
{
	__shared__ long long part[2][ROI_BLOCK];
	const int t = (int)threadIdx.x;
	const long long per = (n + ROI_BLOCK - 1) / ROI_BLOCK;
	const long long lo = (long long)t * per < n ? (long long)t * per : n, hi = lo + per < n ? lo + per : n;
	for (int k = 0; k < 2; ++k) {
		long long s = 0;
		for (long long i = lo; i < hi; ++i) s += cnt[k * (n + 1) + i];
		part[k][t] = s;
	}
	__syncthreads();
	if (t < 2) {
		long long run = 0;
		for (int i = 0; i < ROI_BLOCK; ++i) { const long long c = part[t][i]; part[t][i] = run; run += c; }
		cnt[t * (n + 1) + n] = run;
	}
	__syncthreads();
	for (int k = 0; k < 2; ++k) {
		long long run = part[k][t];
		for (long long i = lo; i < hi; ++i) { const long long c = cnt[k * (n + 1) + i]; cnt[k * (n + 1) + i] = run; run += c; }
	}
}

__global__ __launch_bounds__(ROI_BLOCK) void k_roi_emit(const RoiArgs a)
{
	__shared__ int pre[2][ROI_BLOCK / 64];
	const long long p = (long long)blockIdx.x * ROI_BLOCK + threadIdx.x;
	bool s = false, e = false;
	if (p < a.len) roi_edges(a, p, s, e);
	const unsigned long long ms = ballot(s), me = ballot(e);
	const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63);
	if (l == 0) { pre[0][w] = popc64(ms); pre[1][w] = popc64(me); }
	__syncthreads();
	int bs = 0, be = 0;
	for (int k = 0; k < w; ++k) { bs += pre[0][k]; be += pre[1][k]; }
	const unsigned long long below = l ? (~0ull >> (64 - l)) : 0ull;
	if (s) a.roi_start[a.block_cnt[blockIdx.x] + bs + popc64(ms & below)] = p;
	if (e) a.roi_end[a.block_cnt[a.pos_blocks + 1 + blockIdx.x] + be + popc64(me & below)] = p;
}

// gen_roi_internal's read loop (:479-484): one wave per region.  FILL = false: count (capped at max_reads + 1);
// FILL = true: write the indices of the regions that are yielded (roi_off >= 0).
template <bool FILL>
__global__ __launch_bounds__(64) void k_roi_reads(const RoiArgs a)
{
	const int lane = lane_id();
	for (long long k = blockIdx.x; k < a.n_roi; k += gridDim.x) {
		long long out = 0;
		if (FILL) { out = a.roi_off[k]; if (out < 0) continue; }
		const long long rs = a.roi_start[k], re = a.roi_end[k];
		long long lo = 0, hi = a.n_reads;                     // first read whose inclusive prefix max of stops reaches rs
		while (lo < hi) { const long long m = (lo + hi) >> 1; if (a.pmax_incl[m] >= rs) hi = m; else lo = m + 1; }
		const long long j0 = lo;
		lo = j0; hi = a.n_reads;                              // first read that starts beyond re
		while (lo < hi) { const long long m = (lo + hi) >> 1; if (a.start[m] > re) hi = m; else lo = m + 1; }
		const long long j1 = lo;
		int n = 0;
		for (long long j = j0; j < j1 && n <= a.max_reads; j += 64) {
			const long long i = j + lane;
			const bool ok = i < j1 && !(a.skip && a.skip[i]) && a.stop[i] >= rs;   // overlaps(), :447-450 (start <= re by j1)
			const unsigned long long m = ballot(ok);
			const int rank = n + popc64(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
			if (FILL && ok && rank <= a.max_reads) a.roi_reads[out + rank] = i;   // at most max_reads + 1 are collected (:483)
			n += popc64(m);
		}
		if (!FILL && lane == 0) a.roi_cnt[k] = n > a.max_reads + 1 ? a.max_reads + 1 : n;
	}
}

}  // namespace ihp
