// ihp_common.h -- shared device/host declarations of the HIP implementation.
//
// Execution model used by every kernel in this library: ONE 64-lane wavefront per
// workgroup, one unit of work (region / alignment / contig) per wavefront at a
// time, persistent grid pulling work items from an atomic counter.  All control
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
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p ? 1 : 0); }
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int clz64(unsigned long long m) { return __clzll((long long)m); }
__device__ __forceinline__ int bcast(int v, int src) { return __shfl(v, src, 64); }
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

// Work queue of a persistent grid: eight counters on separate cache lines (item j belongs to shard j & 7).
// A single dequeue word saturates near 90 dequeues/us on this chip, which at ~20k short items per launch
// costs more than the items themselves; a workgroup pulls from its home shard first and steals from the
// others when it runs dry.  Call from lane 0 only.
constexpr int WQ_WORDS = 8 * 16;
__device__ __forceinline__ int wq_next(int *q, int n, int home, unsigned &dead)
{
	for (int t = 0; t < 8; ++t) {
		const int s = (home + t) & 7;
		if ((dead >> s) & 1) continue;
		if (__hip_atomic_load(&q[s * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8 + s < n) {
			const int j = atomicAdd(&q[s * 16], 1) * 8 + s;
			if (j < n) return j;
		}
		dead |= 1u << s;
	}
	return -1;
}

// One alignment job for the ksw2 kernel.
struct AlnJob {
	long long q_off;     // into the query byte array
	long long t_off;     // into the target byte array
	int qlen, tlen;
	int out;             // result slot
	int region;          // -1 for the plain batch API
};

// Device-side event record (host converts to ihp_event and adds the genotype).
struct DevEvent {
	int tstart_rel, tstop_rel;   // relative to ctg.start's region origin (int64 added on host)
	int qstart, qstop;
	unsigned len;
	unsigned char type, status, fallback, pad;
	int cf_offset;
	int ref_support, alt_support, both_found;
	char ref_kmer[32], alt_kmer[32];
};

struct KswParams {
	int m; int sc_mch, sc_mis; int min_sc; int q, e, w, zdrop, flag; int encode_ascii;
	int codes_ok;                             // host: every base code is < m (always so after ASCII encoding)
};

}  // namespace ihp
