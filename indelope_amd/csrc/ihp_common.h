// ihp_common.h -- shared device/host declarations of the HIP implementation.
//
// Execution model used by every kernel in this library: ONE 64-lane wavefront per
// workgroup, one unit of work (region / alignment / contig) per wavefront at a
// time, persistent grid pulling work items from sharded counters (wq_next) or dealt round robin.  All control
// flow around WSYNC() is wave-uniform.  gfx950 only: wave size is hard-coded 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "indelope_hip.h"

namespace ihp {

constexpr int WAVE = 64;
constexpr int MAXLEN = 8192;     // longest contig the assembly kernel will build
constexpr int FILTER_CH = 8;     // bases examined by the per-offset prefilter

// Single-wave workgroup: the barrier itself is free; what matters is the
// s_waitcnt + compiler fence that orders LDS / global scratch traffic between lanes.
#define WSYNC() __syncthreads()
// LDS traffic of one wave is serviced in program order, so between lanes of the SAME wave a compiler
// barrier is all that LDS read-after-write needs; unlike WSYNC it does not wait for global stores.
#define LDS_ORDER() asm volatile("" ::: "memory")

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int clz64(unsigned long long m) { return __clzll((long long)m); }
// Value of lane `src` (wave-uniform index) for every lane: v_readlane, not a trip through the LDS crossbar; the
// result is in an SGPR, so everything computed from it stays on the scalar unit.
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }

// Wave-wide reductions on the DPP network (xor 1, xor 2, 8-lane mirror, 16-lane mirror, then row_bcast 15/31
// fold the four rows into lane 63): six VALU instructions, against six LDS-crossbar round trips for a shuffle loop.
#define IHP_WAVE_REDUCE(NAME, TYPE, OP)                                                                    \
	__device__ __forceinline__ TYPE NAME(TYPE v)                                                           \
	{                                                                                                      \
		int t;                                                                                             \
		asm("s_nop 4\n\t"                                                                                  \
		    OP " %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                   \
		    OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                   \
		    OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                       \
		    OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                            \
		    OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                          \
		    OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"                               \
		    : "=&v"(t) : "v"((int)v));                                                                     \
		return (TYPE)__builtin_amdgcn_readlane(t, 63);                                                     \
	}
IHP_WAVE_REDUCE(wave_min_u32, unsigned, "v_min_u32_dpp")
IHP_WAVE_REDUCE(wave_max_u32, unsigned, "v_max_u32_dpp")
IHP_WAVE_REDUCE(wave_min_i32, int, "v_min_i32_dpp")
IHP_WAVE_REDUCE(wave_max_i32s, int, "v_max_i32_dpp")
IHP_WAVE_REDUCE(wave_sum_i, int, "v_add_u32_dpp")
IHP_WAVE_REDUCE(wave_or_u32, unsigned, "v_or_b32_dpp")

// Inclusive prefix sum over the 64 lanes, the DPP scan: 4-wide sums from three row shifts of the input, then
// row_shr 4 / 8 on the lanes that have such a neighbour, then the row broadcasts.
__device__ __forceinline__ unsigned wave_scan_add(unsigned v)
{
	unsigned t;
	asm("s_nop 4\n\t"
	    "v_mov_b32 %0, %1\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
	    "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
	    : "=&v"(t) : "v"(v));
	return t;
}
// Inclusive prefix maximum of unsigned values over the 64 lanes (same DPP walk; 0 is the identity: zero fill at row starts).
__device__ __forceinline__ unsigned wave_scan_max(unsigned v)
{
	unsigned t;
	asm("s_nop 4\n\t"
	    "v_mov_b32 %0, %1\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
	    "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
	    : "=&v"(t) : "v"(v));
	return t;
}
__device__ __forceinline__ int first_lane_val(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A wave-uniform value that was loaded through the vector memory path sits in a VGPR, and the compiler then
// does all the scalar arithmetic and branching that depends on it on the vector ALU.  uni() moves it to an SGPR.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v)
{
	return ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// contig.nim:44-47 (rule 0) and contig.nim:287-290 (rule 1, the reference's test rule)
__device__ __forceinline__ bool allowed(int rule, uint32_t qsup, uint32_t tsup, long long qreads, long long treads)
{
	if (rule == IHP_ALLOW_SUPPORT)
		return (qsup < 3u && tsup > 3u * qsup) || (tsup < 3u && qsup > 3u * tsup);
	return (qsup < 3u && tsup > 3u * qsup && qreads > 3ll * (long long)qsup) ||
	       (tsup < 3u && qsup > 3u * tsup && treads > 3ll * (long long)tsup);
}

// Work queue of a persistent grid: up to 64 counters on separate cache lines; item j belongs to shard j % S and a
// workgroup serves its home shard (blockIdx % S) only, S = min(64, grid).  One address sustains only ~40 requests
// per microsecond, loads included: a single counter makes 20 000 short items cost more than their work, and eight
// counters with stealing made every wave sweep all eight at the end (65 000 requests per launch of 8192 waves,
// ~0.27 ms).  With one atomic per item plus one per wave, spread over 64 lines, the queue disappears from the
// profile; items are dealt round robin over the shards, so the shards hold equal shares and balancing inside a
// shard (grid / S waves) is enough.  Call from lane 0 only; `dead` is the wave's "my shard is dry" flag.
constexpr int WQ_SHARDS = 64;
constexpr int WQ_WORDS = WQ_SHARDS * 16;
__device__ __forceinline__ int wq_next(int *q, int n, int home, unsigned &dead)
{
	if (dead) return -1;
	const int S = (int)gridDim.x < WQ_SHARDS ? (int)gridDim.x : WQ_SHARDS;
	const int s = home % S;
	const int j = atomicAdd(&q[s * 16], 1) * S + s;
	if (j < n) return j;
	dead = 1u;
	return -1;
}

// One alignment job for the ksw2 kernel.
struct AlnJob {
	long long q_off;     // into the query byte array
	long long t_off;     // into the target byte array
	int qlen, tlen;
	int out;             // result slot
	int region;          // -1 for the plain batch API
	int flags;           // ALN_Q_ACGT: the producer vouches that every query base encodes to A C G T and every target base to A C G T N
	int pad_;
};
constexpr int ALN_Q_ACGT = 1;

// Device-side event record (host converts to ihp_event and adds the genotype).
struct DevEvent {
	int tstart_rel, tstop_rel;   // relative to ctg.start's region origin (int64 added on host)
	int qstart, qstop;
	unsigned len;
	unsigned char type, status, fallback, aligned;
	int cf_offset;
	int ref_support, alt_support, both_found;   // as at indelope.nim:375 (alignment votes when `aligned`)
	char ref_kmer[32], alt_kmer[32];
	int kmer_ref, kmer_alt, kmer_both;          // the k-mer tally itself (indelope.nim:285-311)
	int pad2;
	long long hit_off;                          // into the hit pool: nreads ref positions, then nreads alt positions; -1 none
};

// One event whose k-mer tally found both k-mers in some read (indelope.nim:313): the alignment fallback
// kernel aligns every read of the region for it.
struct FbItem { int job; int ev; };             // AlnJob index, event pool index

struct KswParams {
	int m; int sc_mch, sc_mis; int min_sc; int q, e, w, zdrop, flag; int encode_ascii;
	int codes_ok;                             // host: every base code is < m (always so after ASCII encoding)
};

}  // namespace ihp
