// indelope_hip.hip -- C ABI of include/indelope_hip.h over the gfx950 kernels.
//
// Host code here only moves bytes and launches kernels: uploads the flat batch,
// sizes the per-workgroup scratch, launches assemble -> ksw2 -> tally -> summary
// on one stream, and repacks the slot-indexed device outputs into the flat
// `ihp_batch_out`.  The only arithmetic done on the host is genotype()
// (genotyper.nim:36-47: three fp64 logs per event).  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>
#include "kernels.h"
#include "variants_host.h"

using namespace ihp;

// ------------------------------------------------------------------ context
namespace {

struct Ctx {
	bool ready = false;
	int device = -1;
	int cus = 0;
	int wall_khz = 0;                                      // rate of the device wall clock (wall_clock64) in kHz
	int64_t hbm = 0;
	int max_lds = 65536;
	int comb_static = 4608;                                // static LDS of k_asm_combine3 (hipFuncGetAttributes)
	int comb_static_a = 1536;                              // ... of the first tier's build (room for COMB_MAXC_A contigs)
	int comb_static_w = 4864;                              // ... of the wide build (16-bit supports: the regions of more than 255 reads)
	hipStream_t stream = nullptr;
	char err[512] = "";
};
Ctx g;

int hip_fail(hipError_t e, const char *what, int line)
{
	snprintf(g.err, sizeof(g.err), "%s (line %d): %s", what, line, hipGetErrorString(e));
	return IHP_E_HIP;
}
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x, __LINE__); } while (0)

// HIP's current device is a per-thread setting: ihp_init binds the calling thread, every other host thread that
// drives batches (header: "batches may be driven from several host threads at once") is bound on its first call.
thread_local int tl_device = -1;
int ensure_init()
{
	if (!g.ready) { int rc = ihp_init(0); if (rc) return rc; }
	if (tl_device != g.device) {
		hipError_t e = hipSetDevice(g.device);
		if (e != hipSuccess) return hip_fail(e, "hipSetDevice", __LINE__);
		tl_device = g.device;
	}
	return 0;
}

// test hook (ihp_debug_limits): caps on the device pools so that the overflow paths can be driven by small inputs
long long g_limits[4] = {0, 0, 0, 0};      // CIGAR bump words, event pool entries, hit pool ints, ksw traceback bytes
int g_ksw_status = 0;                      // result of the most recent ksw_extz2_sse call (ihp_ksw_last_status)

// test / diagnostics switches (ihp_debug_set): which of the equivalent paths a batch takes and at what occupancy.  Results
// never depend on them (tests/test_gpu_round3.py runs the parity suite's workloads under every path switch).
struct Knobs {
	int asm_v1 = 0;        // 1: class 1 through the byte-based k_assemble passes only (no packed assembly)
	int no_rich = 0;       // 1: read-rich regions (classes 2-4) stay with the byte-based passes
	int no_hint = 0;       // 1: every combine launch with its full grid whatever the last batch needed
	int no_spec = 0;       // 1: the retry launches are always enqueued (default: left out when the last batch needed none, checked at the wait)
	int ksw_p_cap = 0;     // bytes: caps the traceback scratch per wave of the MAIN ksw2 launch (its jobs that need more go to the roomy launch)
	int fb_p_cap = 0;      // the same for the MAIN launch of the alignment fallback (items that need more go to its roomy launch)
	int comb_waves = 0;    // waves per workgroup of k_asm_combine3 (1, 2, 4; 0 = by the tier's occupancy): wave 0 runs the region, the others share its best_match calls
	int verbose = 0;       // 1: a line on stderr per run with the combine tiers it was launched with
	int tally_rec_cap = 0; // test hook: records k_tally_prep may write (the other jobs with events take k_tally's own header path)
	int tally_minw = 8;    // waves per SIMD k_tally is compiled for (6: 78 VGPRs; 7: 72; 8: 64 and 20 bytes of scratch -- the kernel waits for memory 41 % of its time: 0.86 -> 0.79 ms per 100 000 C2 regions)
	int comb_minw = 5;     // waves per SIMD the first combine tier's build is compiled for (5: 95 VGPRs, 20 regions per CU; 6: 80 and 7: 72 -- with spills: 21 regions per CU run no faster than the 20 of the build without)
	int spec_fail = 0;     // test hook: 1 = a run that left the retry launches out is treated as if a region had needed them
	int tally_pk = 1;      // 0: k_tally reads the ASCII bases even when k_prepack's 2-bit reads are at hand
	int lpt = 1;           // 0: k_asm_combine3 takes its regions in input order (no cost classes, no arena tiers)
	int ksw_pair = 1;      // 0: every alignment through the single sweep (no k_ksw_plan / k_ksw_pair launches)
	int prepack_fast = 2;  // 0: k_prepack for every batch; 1..4: k_prepack_fast<n> (n reads per 16-lane group in flight) when the bases are ASCII and the trim bounds came with the batch
	int fb_skip = 1;       // 0: the two-target sweep computes every slot of query positions on every diagonal (round 5); 1: a slot is left out once it is past the longer target's end
	int fb_duo = 1;        // 0: the alignment fallback runs its two alignments one after the other (ksw_wide.h) instead of in one sweep (ksw_duo.h)
	int asm_waves = 0, asmr_waves = 0, comb_occ = 0, ksw_waves = 0, tally_waves = 0;   // waves per CU (0 = library sizing)
	int v2_arena = 0, v2_pdw = 0;                                                       // LDS sizes of the packed assembly (0 = library sizing)
	int profile = 0;       // 1: kernels sum shader-clock cycles per phase (ihp_batch_profile)
	int strict_ksw = 0;    // 1: ksw_extz2_sse aborts on failure (also IHP_KSW_STRICT=1 in the environment)
};
Knobs g_knob;

// What the last finished batch needed of the combine launches behind the first tier (regions filed under the second and the
// third tier, regions that ran out of room and went to the roomy launch).  A sweep uploads batch after batch of the same
// kind: a launch nobody needed last time is started with a token grid -- an empty launch of workgroups that each ask for
// 25-60 KB of LDS still has to get every one of them scheduled, 50-280 us on the batch's stream in front of k_ksw.
// hist[k]: regions of the last batch whose contigs fit the first-tier arena of HIST_OCC[k] waves per CU and no smaller one
// (the read kernel files them); sig: the shape of that batch (read length, read bases per region): a batch of another
// shape does not use the histogram.
constexpr int HIST_N = 11;
constexpr int HIST_OCC[HIST_N] = {25, 21, 18, 16, 14, 12, 11, 10, 9, 8, 7};     // what 128 LDS granules per CU divide into (see LDS_GRAN)
constexpr int COMB_MAXC_A = 32;                              // contigs the first combine tier's build keeps a table for (V3StateT, asm3_dev.h)
// clean_c / clean_b: batches of this shape in a row, up to now, that filed NO region under the third combine tier / under the second
// and none with more than COMB_MAXC_A contigs.  A first tier that walks the other tiers' lists instead of giving them a
// launch of their own (fold_b / fold_c in ihp_batch_run) rests on that: whatever region IS filed there meets the first tier's
// arena and (short) contig table, is refused, lands on the retry list, and a run that left the retry launches out is then
// repeated in full.  One batch without such a region says little about the next (a sweep whose batches hold one now
// and then paid twice for every batch that followed one without); CLEAN_MIN of them in a row is what the fold waits for.
// clean_r / clean_k: the same for the launches a run leaves out altogether -- batches in a row in which no region took the retry
// route (n_big, n_back) / no job the roomy ksw2 launch (n_kovf).
constexpr int CLEAN_MIN = 3;
struct TierHint { int valid = 0, n_b = 0, n_c = 0, n_big = 0, n_back = 0, n_kovf = 0, regions = 0, sig = 0, n_manyc = 0, wide = 0, clean_c = 0, clean_b = 0, clean_r = 0, clean_k = 0; int hist[HIST_N] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; };
// One hint per batch SHAPE (hint_key: read length, read bases per region, the packed / byte-based path, the parameters that
// decide which launches a run needs), sixteen shapes remembered: a sweep that interleaves batches of different shapes, or
// several host threads with different workloads, keep their plans apart (round 3 had one process-wide hint; only the tier
// histogram was keyed, so a batch that needed the retry route made every other shape re-run, and the other way round).
// Refreshed by whoever confirms a run (sync, fetch, pack, summary, release): after a run that had to be repeated the
// counters are those of the full run, so the next batch of that shape enqueues the launches it needs.
struct HintTable {
	std::mutex mu;
	struct Slot { unsigned long long key = 0; unsigned long long age = 0; TierHint h; } slot[16];
	unsigned long long clock = 0;
	bool get(unsigned long long key, TierHint &out)
	{
		std::lock_guard<std::mutex> lk(mu);
		for (Slot &s : slot) if (s.h.valid && s.key == key) { s.age = ++clock; out = s.h; return true; }
		out = TierHint();
		return false;
	}
	void put(unsigned long long key, const TierHint &h)
	{
		std::lock_guard<std::mutex> lk(mu);
		Slot *best = &slot[0];
		for (Slot &s : slot) {
			if (s.h.valid && s.key == key) { best = &s; break; }
			if (!s.h.valid) { if (best->h.valid) best = &s; }
			else if (best->h.valid && s.age < best->age) best = &s;
		}
		best->key = key; best->h = h; best->h.valid = 1; best->age = ++clock;
	}
	void clear() { std::lock_guard<std::mutex> lk(mu); for (Slot &s : slot) s = Slot(); }
};
HintTable g_hints;
std::atomic<int> g_live_batches{0};

// Device memory comes from a caching pool: a BAM sweep uploads batch after batch of similar shape, and hipMalloc /
// hipFree (which synchronises the device) of ~50 buffers per batch would cost more than the kernels.  Freed blocks are
// kept (up to a byte budget) and handed out again to requests of similar size.
struct DevPool {
	std::mutex mu;
	std::vector<std::pair<void *, size_t>> free_list;
	size_t cached = 0;
	static constexpr size_t BUDGET = (size_t)24 << 30;
	void *get(size_t bytes, size_t *cap) {
		{
			std::lock_guard<std::mutex> l(mu);
			int best = -1;
			for (int i = 0; i < (int)free_list.size(); ++i)
				if (free_list[i].second >= bytes && (best < 0 || free_list[i].second < free_list[best].second)) best = i;
			if (best >= 0 && free_list[best].second <= 2 * bytes + 4096) {
				void *p = free_list[best].first;
				*cap = free_list[best].second;
				cached -= *cap;
				free_list.erase(free_list.begin() + best);
				return p;
			}
		}
		void *p = nullptr;
		if (hipMalloc(&p, bytes) != hipSuccess) {
			clear();                                           // give the cached blocks back and try once more
			if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
		}
		*cap = bytes;
		return p;
	}
	void put(void *p, size_t cap) {
		std::lock_guard<std::mutex> l(mu);
		if (cached + cap > BUDGET) { (void)hipFree(p); return; }
		free_list.push_back({p, cap});
		cached += cap;
	}
	void clear() {
		std::lock_guard<std::mutex> l(mu);
		for (auto &e : free_list) (void)hipFree(e.first);
		free_list.clear(); cached = 0;
	}
};
DevPool g_pool;

// Every device-resident batch runs on a stream of its own, so batches in flight from different host threads (one
// uploading, one running, one fetching) overlap on the copy engines and the CUs.  Streams are reused.
struct StreamCache {
	std::mutex mu;
	std::vector<hipStream_t> free_list;
	hipStream_t get() {
		{
			std::lock_guard<std::mutex> l(mu);
			if (!free_list.empty()) { hipStream_t s = free_list.back(); free_list.pop_back(); return s; }
		}
		hipStream_t s = nullptr;
		if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
		return s;
	}
	// Only a few are kept: the runtime spreads the streams that EXIST over its hardware queues (the least used queue gets
	// the next stream), so a pile of idle cached streams makes two live ones share a queue sooner.
	void put(hipStream_t s) {
		{
			std::lock_guard<std::mutex> l(mu);
			if (free_list.size() < 4) { free_list.push_back(s); return; }
		}
		(void)hipStreamDestroy(s);
	}
	void clear() {
		std::lock_guard<std::mutex> l(mu);
		for (auto s : free_list) (void)hipStreamDestroy(s);
		free_list.clear();
	}
};
StreamCache g_streams;

// A run's counters, overflow flags and stamps are left by its last kernel in a 256-byte block of page-locked host
// memory (k_summary writes it over PCIe), so that ihp_batch_sync / fetch / profile read them without a device copy.
constexpr int REPORT_INTS = 64;
constexpr int REPORT_BLOCK = 128;                            // ints per block: the report, then the five totals of k_pack_scan (int64 each, from int 64)
struct ReportPool {
	std::mutex mu;
	std::vector<int *> free_list, pages;
	int *get() {
		std::lock_guard<std::mutex> l(mu);
		if (free_list.empty()) {
			void *pg = nullptr;
			if (hipHostMalloc(&pg, 4096, hipHostMallocDefault) != hipSuccess) return nullptr;
			pages.push_back((int *)pg);
			for (int k = 0; k < 4096 / (int)(sizeof(int) * REPORT_BLOCK); ++k) free_list.push_back((int *)pg + k * REPORT_BLOCK);
		}
		int *r = free_list.back(); free_list.pop_back();
		memset(r, 0, sizeof(int) * REPORT_BLOCK);
		return r;
	}
	void put(int *r) { std::lock_guard<std::mutex> l(mu); free_list.push_back(r); }
	void clear() {
		std::lock_guard<std::mutex> l(mu);
		// a live batch still points into its page (k_summary writes there, sync / fetch read it): the pages stay until no batch does
		if (g_live_batches.load() > 0) return;
		for (auto pg : pages) (void)hipHostFree(pg);
		pages.clear(); free_list.clear();
	}
};
ReportPool g_reports;

// Page-locked host slabs (results of ihp_batch_fetch, staging blocks of ihp_batch_upload*): pinning is expensive, so freed
// slabs are kept and reused.  64-byte header + payload.
struct SlabHdr { uint64_t magic; size_t cap; uint64_t pad[6]; };
static_assert(sizeof(SlabHdr) == 64, "slab header");
constexpr uint64_t SLAB_MAGIC = 0x49485053'4c414231ull;
struct SlabCache {
	struct Ent { void *base; size_t cap; unsigned long long tick; };
	std::mutex mu;
	std::vector<Ent> free_list;
	unsigned long long clock = 0;
	size_t bytes_kept = 0;
	void *get(size_t bytes) {
		{
			std::lock_guard<std::mutex> l(mu);
			int best = -1;
			for (int i = 0; i < (int)free_list.size(); ++i)
				if (free_list[i].cap >= bytes && (best < 0 || free_list[i].cap < free_list[best].cap)) best = i;
			if (best >= 0 && free_list[best].cap <= 4 * bytes + (1u << 20)) {
				void *p = free_list[best].base;
				bytes_kept -= free_list[best].cap;
				free_list.erase(free_list.begin() + best);
				return p;
			}
		}
		const size_t cap = bytes + bytes / 4 + 4096;
		void *p = nullptr;
		if (hipHostMalloc(&p, cap + sizeof(SlabHdr), hipHostMallocDefault) != hipSuccess) return nullptr;
		SlabHdr *h = (SlabHdr *)p;
		h->magic = SLAB_MAGIC; h->cap = cap;
		return p;
	}
	// Freed blocks are kept -- result slabs and staging blocks of every batch in flight, of a few sizes -- up to 64 blocks and
	// 4 GB; beyond that the one that has been lying here longest goes (hipHostFree waits for the device: an eviction policy
	// that threw out the smallest block first made every batch of a sweep allocate and free its staging block again).
	void put(void *base) {
		std::vector<void *> drop;
		{
			std::lock_guard<std::mutex> l(mu);
			free_list.push_back({base, ((SlabHdr *)base)->cap, ++clock});
			bytes_kept += free_list.back().cap;
			while (free_list.size() > 64 || (bytes_kept > ((size_t)4 << 30) && free_list.size() > 1)) {
				int w = 0;
				for (int i = 1; i < (int)free_list.size(); ++i) if (free_list[i].tick < free_list[w].tick) w = i;
				drop.push_back(free_list[w].base);
				bytes_kept -= free_list[w].cap;
				free_list.erase(free_list.begin() + w);
			}
		}
		for (void *p : drop) (void)hipHostFree(p);
	}
	void clear() {
		std::lock_guard<std::mutex> l(mu);
		for (auto &e : free_list) (void)hipHostFree(e.base);
		free_list.clear();
		bytes_kept = 0;
	}
};
SlabCache g_slabs;

// device buffer; returned to the pool on scope exit / batch free
struct DBuf {
	void *p = nullptr; size_t n = 0, cap = 0;
	bool view_ = false;                                      // part of another buffer (the batch's input slab): not given back on its own
	void view(void *base, size_t off, size_t bytes) { release(); p = (char *)base + off; n = bytes; cap = 0; view_ = true; }
	int alloc(size_t bytes) {
		release();
		n = bytes;
		p = g_pool.get(bytes + 64, &cap);                    // +64: kernels read whole dwords at the tail of byte arrays
		if (!p) { n = cap = 0; snprintf(g.err, sizeof(g.err), "hipMalloc of %zu bytes failed", bytes + 64); return IHP_E_NOMEM; }
		return 0;
	}
	int upload(const void *src, size_t bytes, hipStream_t s) {
		int rc = alloc(bytes);
		if (rc) return rc;
		if (bytes) HIPC(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
		return 0;
	}
	int zero(hipStream_t s) { if (n) HIPC(hipMemsetAsync(p, 0, n, s)); return 0; }
	template <class T> T *as() const { return (T *)p; }
	void release() { if (p && !view_) g_pool.put(p, cap); p = nullptr; n = cap = 0; view_ = false; }
	~DBuf() { release(); }
	DBuf() = default;
	DBuf(const DBuf &) = delete;
	DBuf &operator=(const DBuf &) = delete;
};

int grid_for(int items, int waves_per_cu)
{
	long long g_ = (long long)g.cus * waves_per_cu;
	if (g_ > items) g_ = items;
	if (g_ < 1) g_ = 1;
	return (int)g_;
}

int ksw_mode(const KswParams &P)
{
	const int right = (P.flag & KSW_EZ_RIGHT) ? 1 : 0;
	if (P.flag & (KSW_EZ_SCORE_ONLY | KSW_EZ_GENERIC_SC | KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP)) return 2;   // the LDS sweep does every flag
	if (P.w < 0 || P.w > 62) {
		KswParams Q = P; Q.w = 0;
		return (!right && ksw_narrow_ok(Q)) ? 5 : 2;             // 5: ring sweep per job where it fits, LDS sweep otherwise
	}
	return (ksw_narrow_ok(P) && P.codes_ok) ? 3 + right : right;
}

int g_last_ksw_mode = -1;

size_t ksw_mode_lds(int mode, int qlen, int tlen)
{
	if (mode == 5) return std::max(ksw_lds_bytes(qlen, tlen), std::max(ksw_wide_lds_bytes<3>(qlen, tlen), ksw_wide_lds_bytes<6>(qlen, tlen)));
	return mode >= 3 ? ksw_narrow_lds_bytes(qlen, tlen) : mode != 2 ? ksw_fast_lds_bytes(qlen, tlen) : ksw_lds_bytes(qlen, tlen);
}

template <class... Args> void launch_ksw(int mode, dim3 grid, size_t lds, hipStream_t s, const KswArgs &a)
{
	if (mode == 0) hipLaunchKernelGGL(k_ksw<0>, grid, dim3(64), lds, s, a);
	else if (mode == 1) hipLaunchKernelGGL(k_ksw<1>, grid, dim3(64), lds, s, a);
	else if (mode == 3) hipLaunchKernelGGL(k_ksw<3>, grid, dim3(64), lds, s, a);
	else if (mode == 4) hipLaunchKernelGGL(k_ksw<4>, grid, dim3(64), lds, s, a);
	else if (mode == 5) hipLaunchKernelGGL(k_ksw<5>, grid, dim3(64), lds, s, a);
	else hipLaunchKernelGGL(k_ksw<2>, grid, dim3(64), lds, s, a);
}

// The production ksw2 stage as three launches: k_ksw_plan pairs the jobs of equal contig length the pair sweep can take
// (ksw_pair.h), the single sweep walks the rest through in_list, k_ksw_pair takes the pairs.  `a` is the single launch's
// argument block (in_list null); the plan's lists live in `plan` (ints: order[n] | singles[n] | pairs[2n]), its two counts
// at `cnt`, the pair launch's work queue at `wq_pair`; p_pair / ct_pair: traceback and CIGAR scratch of the pair launch
// (p_cap_pair, a.cig_cap per workgroup).  Returns false when these parameters are not the pair sweep's (nothing launched).
static bool ksw_pair_wanted(const KswParams &P, int mode) { return g_knob.ksw_pair && mode == 3 && ksw_pair_ok(P); }
// plan: ints [n_cap: ranks | n_cap: singles | 2 n_cap: pairs | PLAN_KEYS: first pair of a length]; zero: ints that are zero when the run starts
// [PLAN_KEYS: jobs per contig length | pairs | singles | finished workgroups of the count]
struct PlanLists { int *count, *n_pairs, *n_singles, *done, *rank, *singles, *pbase; int2 *pairs; };
static PlanLists plan_lists(int *plan, int n_cap, int *zero)
{
	PlanLists L;
	L.count = zero; L.n_pairs = zero + PLAN_KEYS; L.n_singles = zero + PLAN_KEYS + 1; L.done = zero + PLAN_KEYS + 2;
	L.rank = plan; L.singles = plan + n_cap; L.pairs = (int2 *)(plan + 2 * (size_t)n_cap + (n_cap & 1));
	L.pbase = plan + 4 * (size_t)n_cap + 8;
	return L;
}
static int launch_ksw_planned(dim3 grid, size_t lds_single, size_t lds_pair, hipStream_t s, const KswArgs &a, int *plan, int n_cap, int *zero,
                              int *wq_pair, uint8_t *p_pair, size_t p_cap_pair, uint32_t *ct_pair,
                              hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr)
{
	const PlanLists L = plan_lists(plan, n_cap, zero);
	KswPlanArgs pl;
	pl.jobs = a.jobs; pl.n_jobs = a.n_jobs; pl.n_jobs_host = a.n_jobs_host; pl.P = a.P; pl.pair_on = 1;
	pl.lds_budget = (int)lds_pair - 64; pl.p_cap = p_cap_pair;
	pl.count = L.count; pl.n_pairs = L.n_pairs; pl.n_singles = L.n_singles;
	pl.t_start = a.t_start;
	pl.rank = L.rank; pl.singles = L.singles; pl.pairs = L.pairs; pl.done = L.done; pl.pbase = L.pbase;
	const dim3 pg(std::max(1, std::min(2 * g.cus, (n_cap + 255) / 256)));
	hipLaunchKernelGGL(k_ksw_plan_count, pg, dim3(256), 0, s, pl);
	hipLaunchKernelGGL(k_ksw_plan_place, pg, dim3(256), 0, s, pl);
	// the jobs without a partner: a few per cent, each a serial chain of ~0.1 ms -- beside the pairs, not in front of them
	KswArgs x = a;
	x.t_start = nullptr; x.in_list = pl.singles; x.n_jobs = pl.n_singles;
	if (side) {
		HIPC(hipEventRecord(ev_fork, s));
		HIPC(hipStreamWaitEvent(side, ev_fork, 0));
		launch_ksw(3, grid, lds_single, side, x);
		HIPC(hipEventRecord(ev_join, side));
	} else launch_ksw(3, grid, lds_single, s, x);
	KswArgs y = a;
	y.t_start = nullptr; y.in_list = nullptr; y.pairs = pl.pairs; y.n_jobs = pl.n_pairs; y.ovf_list = nullptr; y.ovf_n = nullptr;
	y.lds_budget = (int)lds_pair - 64; y.p_scratch = p_pair; y.p_cap = p_cap_pair; y.cig_tmp = ct_pair; y.work_counter = wq_pair;
	hipLaunchKernelGGL(k_ksw_pair, grid, dim3(64), lds_pair, s, y);
	if (side) HIPC(hipStreamWaitEvent(s, ev_join, 0));
	return 0;
}
static size_t ksw_plan_ints(long long n_cap) { return 4 * (size_t)n_cap + 8 + PLAN_KEYS; }
constexpr int PLAN_ZERO_INTS = PLAN_KEYS + 16;

KswParams make_ksw_params(int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag, int ascii)
{
	KswParams P;
	P.m = m; P.sc_mch = mat[0]; P.sc_mis = mat[1];
	int mn = mat[1];
	for (int t = 1; t < m * m; ++t) mn = mn < mat[t] ? mn : mat[t];      // ksw2_extz2_sse.c:167-170
	P.min_sc = mn; P.q = q; P.e = e; P.w = w; P.zdrop = zdrop; P.flag = flag; P.encode_ascii = ascii; P.codes_ok = ascii;
	return P;
}

}  // namespace

// ------------------------------------------------------------------ basics
extern "C" const char *ihp_strerror(int code)
{
	switch (code) {
	case IHP_OK: return "ok";
	case IHP_E_NODEVICE: return "no usable gfx950 device";
	case IHP_E_HIP: return "HIP runtime error";
	case IHP_E_ARG: return "bad argument";
	case IHP_E_NOMEM: return "out of memory";
	case IHP_E_CAPACITY: return "capacity exceeded";
	case IHP_E_UNSUPPORTED: return "not supported on the GPU path";
	}
	return "unknown error";
}

extern "C" const char *ihp_last_hip_error(void) { return g.err; }
extern "C" const char *ihp_version(void) { return "indelope_hip 0.1 (gfx950)"; }

static void slab_cache_clear();
static void report_pool_clear();
extern "C" int ihp_init(int device)
{
	if (g.ready && g.device == device) { tl_device = -1; return ensure_init(); }
	// Every batch in flight has two streams of its own, and the HIP runtime multiplexes all streams of a process onto
	// GPU_MAX_HW_QUEUES hardware queues (default 4): with more than two batches about, two launch chains that should
	// overlap land on one queue every few runs and run one after the other (C3 / C5 behind the e2e leg of bench.py: 4.4 ->
	// 3.8 M and 2.6 -> 2.1 M regions/s in two runs of five; none in five with 16 queues).  The variable is read when the
	// runtime starts, so it is the CALLER's to set before its first HIP call (include/indelope_hip.h says so; the Python
	// package does it at import).  The library no longer calls setenv(): that is not safe beside a getenv() in another host
	// thread, and it has no effect once the runtime is up.  A process that did not set it gets a one-line note under "verbose".
	if (g_knob.verbose && !getenv("GPU_MAX_HW_QUEUES")) fprintf(stderr, "[ihp] GPU_MAX_HW_QUEUES is not set: the runtime's default of 4 hardware queues lets launch chains of different batches share a queue\n");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { snprintf(g.err, sizeof(g.err), "no HIP device"); return IHP_E_NODEVICE; }
	if (device < 0 || device >= n) return IHP_E_ARG;
	HIPC(hipSetDevice(device));
	hipDeviceProp_t pr;
	HIPC(hipGetDeviceProperties(&pr, device));
	if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) {
		snprintf(g.err, sizeof(g.err), "device %d is %s; this library is built for gfx950 only", device, pr.gcnArchName);
		return IHP_E_NODEVICE;
	}
	if (g.ready && g.device != device) {
		// pooled blocks, streams and pinned slabs belong to the old device
		(void)hipSetDevice(g.device);
		if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
		slab_cache_clear(); g_pool.clear(); g_streams.clear(); report_pool_clear();
		g.ready = false;
		HIPC(hipSetDevice(device));
	}
	if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
	HIPC(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
	tl_device = device;
	g.device = device; g.cus = pr.multiProcessorCount; g.hbm = (int64_t)pr.totalGlobalMem;
	g.max_lds = (int)pr.sharedMemPerBlock;
	if (hipDeviceGetAttribute(&g.wall_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) g.wall_khz = 0;
	if (g.max_lds > 65536) {
		// opt in to the full 160 KiB LDS for the ksw2 kernel's dynamic region
		(void)hipFuncSetAttribute((const void *)k_assemble<256, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 16384);
		(void)hipFuncSetAttribute((const void *)k_tally<6>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 4096);
		(void)hipFuncSetAttribute((const void *)k_tally<7>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 4096);
		(void)hipFuncSetAttribute((const void *)k_tally<8>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 4096);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<5, false, COMB_MAXC_A>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<6, false, COMB_MAXC_A>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<7, false, COMB_MAXC_A>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<4, true, COMB_MAXC_A>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		(void)hipFuncSetAttribute((const void *)k_asm_combine3<4, false, V3_MAXC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 8192);
		{
			hipFuncAttributes fa;
			if (hipFuncGetAttributes(&fa, (const void *)k_asm_combine3<5, false>) == hipSuccess && fa.sharedSizeBytes > 0) g.comb_static = (int)fa.sharedSizeBytes;
			if (hipFuncGetAttributes(&fa, (const void *)k_asm_combine3<5, false, COMB_MAXC_A>) == hipSuccess && fa.sharedSizeBytes > 0) g.comb_static_a = (int)fa.sharedSizeBytes;
			if (hipFuncGetAttributes(&fa, (const void *)k_asm_combine3<4, false, V3_MAXC, true>) == hipSuccess && fa.sharedSizeBytes > 0) g.comb_static_w = (int)fa.sharedSizeBytes;
		}
		(void)hipFuncSetAttribute((const void *)k_asm_reads<8>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 2048);
		(void)hipFuncSetAttribute((const void *)k_ksw<0>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_ksw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_ksw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_fallback, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_ksw<3>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_ksw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
		(void)hipFuncSetAttribute((const void *)k_ksw<5>, hipFuncAttributeMaxDynamicSharedMemorySize, g.max_lds - 1024);
	}
	g.ready = true;
	return 0;
}

extern "C" int ihp_device_info(int *cu_count, int *wave_size, int64_t *hbm_bytes)
{
	int rc = ensure_init();
	if (rc) return rc;
	if (cu_count) *cu_count = g.cus;
	if (wave_size) *wave_size = 64;
	if (hbm_bytes) *hbm_bytes = g.hbm;
	return 0;
}

extern "C" void ihp_shutdown(void)
{
	if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
	slab_cache_clear();
	report_pool_clear();
	g_pool.clear();
	g_streams.clear();
	g.ready = false; g.device = -1; tl_device = -1;
}

extern "C" int ihp_debug_limits(const int64_t limits[4])
{
	for (int k = 0; k < 4; ++k) g_limits[k] = limits ? (long long)limits[k] : 0;
	return 0;
}

extern "C" int ihp_debug_set(const char *key, int64_t value)
{
	if (!key) { g_knob = Knobs(); g_hints.clear(); return 0; }
	struct { const char *name; int *field; } tab[] = {
		{"asm_v1", &g_knob.asm_v1}, {"no_rich", &g_knob.no_rich}, {"no_hint", &g_knob.no_hint}, {"no_spec", &g_knob.no_spec}, {"spec_fail", &g_knob.spec_fail}, {"comb_minw", &g_knob.comb_minw}, {"tally_minw", &g_knob.tally_minw}, {"tally_rec_cap", &g_knob.tally_rec_cap}, {"verbose", &g_knob.verbose}, {"comb_waves", &g_knob.comb_waves}, {"ksw_p_cap", &g_knob.ksw_p_cap}, {"fb_p_cap", &g_knob.fb_p_cap}, {"tally_pk", &g_knob.tally_pk}, {"lpt", &g_knob.lpt}, {"ksw_pair", &g_knob.ksw_pair}, {"fb_duo", &g_knob.fb_duo}, {"fb_skip", &g_knob.fb_skip}, {"prepack_fast", &g_knob.prepack_fast},
		{"asm_waves", &g_knob.asm_waves}, {"asmr_waves", &g_knob.asmr_waves}, {"comb_occ", &g_knob.comb_occ},
		{"ksw_waves", &g_knob.ksw_waves}, {"tally_waves", &g_knob.tally_waves}, {"v2_arena", &g_knob.v2_arena},
		{"v2_pdw", &g_knob.v2_pdw}, {"profile", &g_knob.profile}, {"strict_ksw", &g_knob.strict_ksw},
	};
	for (auto &e : tab) if (!strcmp(e.name, key)) { *e.field = (int)value; return 0; }
	return IHP_E_ARG;
}

extern "C" void ihp_encode(const uint8_t *dna, int64_t n, uint8_t *out)
{                                                       // ksw2.nim:127-132 (byte LUT, host helper)
	for (int64_t i = 0; i < n; ++i) {
		switch (dna[i]) {
		case 'A': case 'a': out[i] = 0; break;
		case 'C': case 'c': out[i] = 1; break;
		case 'G': case 'g': out[i] = 2; break;
		case 'T': case 't': out[i] = 3; break;
		default: out[i] = 4;
		}
	}
}

extern "C" void ihp_matrix(int8_t match, int8_t mismatch, int8_t out25[25])
{                                                       // ksw2.nim:135-140
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 5; ++j) out25[i * 5 + j] = (i == 4 || j == 4) ? 0 : (i == j ? match : mismatch);
}

extern "C" void ihp_params_default(ihp_params *p)
{
	memset(p, 0, sizeof(*p));
	p->struct_size = (int32_t)sizeof(*p);
	p->min_overlap_pct = 0.88; p->min_mapq_assemble = 20; p->min_mapq_stop = 5; p->min_mapq_tally = 10;
	p->trim_min_qual = 15; p->combine_min_support = 3; p->combine_min_overlap = 65; p->max_mismatch = 0;
	p->max_pre_contigs = 20; p->min_ctg_len = 74; p->min_reads = 4; p->min_event_len = 4;
	p->K = 27; p->max_events = 4; p->ref_pad = 50;
	p->match = 1; p->mismatch = -2; p->gap_open = 4; p->gap_ext = 1;
	p->bw = 50; p->zdrop = 400; p->ksw_flag = 0;
	p->error = 1e-3;
	p->fallback = 1;
	p->fb_match = 1; p->fb_mismatch = -2; p->fb_gap_open = 5; p->fb_gap_ext = 1;   // indelope.nim:318-319
	p->fb_bw = -1; p->fb_zdrop = -1; p->fb_flag = 0;                                // ksw2.nim:159
}

// ------------------------------------------------------------- genotyper.nim
extern "C" int ihp_genotype(int64_t r, int64_t a, double error, ihp_genotype_t *out)
{                                                       // genotyper.nim:36-47
	if (!out) return IHP_E_ARG;
	const double log2_ = log(2.0);
	const double total = (double)(r + a);
	out->gt = IHP_GT_HOM_REF; out->_pad = 0;
	out->gl[0] = out->gl[1] = out->gl[2] = 0.0;
	if (total == 0) { out->gt = IHP_GT_UNKNOWN; return 0; }
	for (int G = 0; G <= 2; ++G) {
		const double gd = (double)G, hd = (double)(2 - G);
		out->gl[G] = -total * log2_ + (double)r * log(gd * error + hd * (1 - error))
		             + (double)a * log(gd * (1 - error) + hd * error);
		if (out->gl[G] > out->gl[out->gt]) out->gt = G;
	}
	return 0;
}

extern "C" double ihp_genotype_qual(const ihp_genotype_t *g_)
{                                                       // genotyper.nim:22-29
	if (g_->gt == IHP_GT_HOM_REF) return g_->gl[0] - fmax(g_->gl[1], g_->gl[2]);
	if (g_->gt == IHP_GT_HET) return g_->gl[1] - fmax(g_->gl[0], g_->gl[2]);
	if (g_->gt == IHP_GT_HOM_ALT) return g_->gl[2] - fmax(g_->gl[0], g_->gl[1]);
	return 0;
}

// ----------------------------------------------------------------- ksw2 batch
// diagnostics: which k_ksw<MODE> the last ksw_extz2_sse / ihp_ksw_extz2_batch / ihp_batch_run call used
extern "C" int ihp_debug_last_ksw_mode(void) { return g_last_ksw_mode; }
// diagnostics: how many PAIRS of alignments the last ihp_ksw_extz2_batch call ran two to a wavefront (ksw_pair.h)
static std::atomic<int> g_last_ksw_pairs{0};
extern "C" int ihp_debug_last_ksw_pairs(void) { return g_last_ksw_pairs.load(); }

static int codes_below(const uint8_t *s, size_t n, int m)
{
	uint8_t mx = 0;
	for (size_t i = 0; i < n; ++i) mx = mx > s[i] ? mx : s[i];
	return (int)mx < m;
}

// The per-region path needs the CIGAR and the exact maximum of every contig alignment (the event iterators and the
// tally start from them): flags that drop either are refused there.  The stand-alone alignment entry points
// (ksw_extz2_sse, ihp_ksw_extz2_batch) take every flag of ksw2.h:8-16 (the LDS sweep, MODE 2, does score-only, generic
// scoring and the approximate maximum; the splice flags mean nothing to ksw_extz2_sse and are ignored as there).
static int ksw_flags_supported(int flag)
{
	return !(flag & (KSW_EZ_SCORE_ONLY | KSW_EZ_GENERIC_SC | KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP));
}
static int ksw_flags_supported_standalone(int flag, int m)
{
	return !(flag & KSW_EZ_GENERIC_SC) || m <= 8;             // the matrix travels in the launch arguments: 64 entries
}

// Run `n` jobs; qbase/tbase already on the device.  Results in host vectors.
static int run_ksw_jobs(const std::vector<AlnJob> &jobs, const uint8_t *d_q, const uint8_t *d_t, const KswParams &P,
                        std::vector<KswOut> &ez, std::vector<long long> &coff, std::vector<uint32_t> &pool, const int8_t *mat = nullptr)
{
	const int n = (int)jobs.size();
	ez.assign(n, KswOut());
	coff.assign(n, -1);
	pool.clear();
	if (n == 0) return 0;
	size_t lds_need = 0, p_need = 0; long long cig_bound = 0; int cig_cap = 0;
	for (const AlnJob &j : jobs) {
		if (j.qlen <= 0 || j.tlen <= 0) continue;
		lds_need = std::max(lds_need, ksw_mode(P) == 2 ? ksw_lds_bytes(j.qlen, j.tlen) : (P.w >= 0 && P.w <= 62) ? ksw_narrow_lds_bytes(j.qlen, j.tlen)
		                    : std::max(ksw_lds_bytes(j.qlen, j.tlen), ksw_wide_ok<3>(P, j.qlen, j.tlen) ? ksw_wide_lds_bytes<3>(j.qlen, j.tlen)
		                               : ksw_wide_ok<6>(P, j.qlen, j.tlen) ? ksw_wide_lds_bytes<6>(j.qlen, j.tlen) : (size_t)0));
		int w = P.w < 0 ? std::max(j.qlen, j.tlen) : P.w;
		int nc = (std::min(std::min(j.qlen, j.tlen), w + 1) + 15) / 16 + 1;
		p_need = std::max(p_need, std::max(((size_t)(j.qlen + j.tlen - 1) * nc + 1) * 16, ksw_narrow_p_bytes(j.qlen, j.tlen)));
		cig_bound += j.qlen + j.tlen;
		cig_cap = std::max(cig_cap, j.qlen + j.tlen);
	}
	if ((long long)lds_need > g.max_lds - 2048) { snprintf(g.err, sizeof(g.err), "alignment needs %zu B of LDS", lds_need); return IHP_E_CAPACITY; }
	const int mode = ksw_mode(P);
	const int grid = grid_for(n, 32);
	DBuf d_jobs, d_p, d_ct, d_ez, d_coff, d_pool, d_misc;
	int rc;
	if ((rc = d_jobs.upload(jobs.data(), sizeof(AlnJob) * n, g.stream))) return rc;
	if ((rc = d_p.alloc((p_need + 64) * grid))) return rc;
	if ((rc = d_ct.alloc(sizeof(uint32_t) * (size_t)(cig_cap + 4) * grid))) return rc;
	if ((rc = d_ez.alloc(sizeof(KswOut) * n))) return rc;
	if ((rc = d_coff.alloc(sizeof(long long) * n))) return rc;
	const long long fixed_words = (long long)n * CIG_SLOT;
	if ((rc = d_pool.alloc(sizeof(uint32_t) * (size_t)(cig_bound + 4 + fixed_words)))) return rc;
	if ((rc = d_misc.alloc(64 + sizeof(int) * (WQ_WORDS * 2 + PLAN_ZERO_INTS)))) return rc;  // [0] cursor u64, [2..4] overflow, [16..] two work queues, then the plan's zero block
	if ((rc = d_misc.zero(g.stream))) return rc;
	KswArgs a;
	a.jobs = d_jobs.as<AlnJob>(); a.n_jobs = nullptr; a.n_jobs_host = n;
	a.qbase = d_q; a.tbase = d_t; a.P = P; a.lds_budget = (int)lds_need;
	a.p_scratch = d_p.as<uint8_t>(); a.p_cap = p_need + 64;
	a.cig_tmp = d_ct.as<uint32_t>(); a.cig_cap = cig_cap + 4;
	a.ez = d_ez.as<KswOut>(); a.cig_off = d_coff.as<long long>();
	a.cig_pool = d_pool.as<uint32_t>(); a.cig_cursor = d_misc.as<unsigned long long>();
	a.cig_bump_cap = cig_bound + 4; a.cig_pool_cap = cig_bound + 4 + fixed_words;
	a.overflow = d_misc.as<int>() + 2; a.work_counter = d_misc.as<int>() + 16; a.prof = nullptr; a.t_start = nullptr;
	a.in_list = nullptr; a.pairs = nullptr; a.ovf_list = nullptr; a.ovf_n = nullptr;   // (this entry point sizes LDS and scratch for its longest pair)
	a.gm = 0; memset(a.gmat, 0, sizeof(a.gmat));
	if ((P.flag & KSW_EZ_GENERIC_SC) && mat && P.m <= 8) { a.gm = P.m; for (int i = 0; i < P.m * P.m; ++i) a.gmat[i] = mat[i]; }
	g_last_ksw_mode = mode;
	DBuf d_plan, d_pp, d_pct;
	if (ksw_pair_wanted(P, mode) && n >= 2) {
		// two alignments per wavefront where the jobs allow it (ksw_pair.h); the rest through the single sweep
		size_t lds_pair = 0, p_pair = 0;
		for (const AlnJob &j : jobs)
			if ((j.flags & ALN_Q_ACGT) && ksw_pair_job_ok(P, j.qlen, j.tlen)) {
				lds_pair = std::max(lds_pair, 2 * ksw_pair_lds_share(j.qlen, j.tlen)); p_pair = std::max(p_pair, ksw_pair_p_bytes(j.qlen, P.w));
			}
		lds_pair = std::min(lds_pair + 64, (size_t)g.max_lds - 2048);
		if ((rc = d_plan.alloc(sizeof(int) * ksw_plan_ints(n)))) return rc;
		if ((rc = d_pp.alloc((p_pair + 64) * grid))) return rc;
		if ((rc = d_pct.alloc(sizeof(uint32_t) * (size_t)(cig_cap + 4) * grid))) return rc;
		launch_ksw_planned(dim3(grid), lds_need + 64, lds_pair + 64, g.stream, a, d_plan.as<int>(), n, d_misc.as<int>() + 16 + 2 * WQ_WORDS,
		                   d_misc.as<int>() + 16 + WQ_WORDS, d_pp.as<uint8_t>(), p_pair + 64, d_pct.as<uint32_t>());
	} else launch_ksw(g_last_ksw_mode, dim3(grid), lds_need + 64, g.stream, a);
	HIPC(hipGetLastError());
	long long misc[8];
	int plan_counts[2] = {0, 0};
	HIPC(hipMemcpyAsync(ez.data(), d_ez.p, sizeof(KswOut) * n, hipMemcpyDeviceToHost, g.stream));
	HIPC(hipMemcpyAsync(coff.data(), d_coff.p, sizeof(long long) * n, hipMemcpyDeviceToHost, g.stream));
	HIPC(hipMemcpyAsync(misc, d_misc.p, 64, hipMemcpyDeviceToHost, g.stream));
	if (d_plan.p) HIPC(hipMemcpyAsync(plan_counts, d_misc.as<int>() + 16 + 2 * WQ_WORDS + PLAN_KEYS, 8, hipMemcpyDeviceToHost, g.stream));
	HIPC(hipStreamSynchronize(g.stream));
	const long long used = misc[0];
	const int *ov = (const int *)misc + 2;
	g_last_ksw_pairs = plan_counts[0];
	if (ov[0] || ov[1]) { snprintf(g.err, sizeof(g.err), "ksw2 kernel capacity overflow (%d,%d)", ov[0], ov[1]); return IHP_E_CAPACITY; }
	(void)used;
	pool.resize((size_t)a.cig_pool_cap);
	HIPC(hipMemcpy(pool.data(), d_pool.p, sizeof(uint32_t) * pool.size(), hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int ihp_ksw_extz2_batch(int32_t n, const uint8_t *queries, const int64_t *q_off,
                                   const uint8_t *targets, const int64_t *t_off,
                                   int8_t m, const int8_t *mat, int8_t q, int8_t e,
                                   int w, int zdrop, int flag,
                                   ihp_ez *ez, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_off)
{
	if (n < 0 || !q_off || !t_off || !mat || !ez || !cigar_off || (n && (!queries || !targets))) return IHP_E_ARG;
	if (!ksw_flags_supported_standalone(flag, m)) return IHP_E_UNSUPPORTED;
	int rc = ensure_init();
	if (rc) return rc;
	cigar_off[0] = 0;
	if (n == 0) return 0;
	std::vector<AlnJob> jobs(n);
	for (int i = 0; i < n; ++i) {
		jobs[i].q_off = q_off[i]; jobs[i].t_off = t_off[i];
		jobs[i].qlen = (int)(q_off[i + 1] - q_off[i]); jobs[i].tlen = (int)(t_off[i + 1] - t_off[i]);
		jobs[i].out = i; jobs[i].region = -1; jobs[i].pad_ = 0;
		// what the pair sweep needs to know (ksw_pair.h): no wildcard in the query, nothing above the wildcard in the target
		jobs[i].flags = (m == 5 && codes_below(queries + q_off[i], (size_t)jobs[i].qlen, 4) && codes_below(targets + t_off[i], (size_t)jobs[i].tlen, 5)) ? ALN_Q_ACGT : 0;
	}
	DBuf d_q, d_t;
	if ((rc = d_q.upload(queries, (size_t)q_off[n], g.stream))) return rc;
	if ((rc = d_t.upload(targets, (size_t)t_off[n], g.stream))) return rc;
	std::vector<KswOut> out; std::vector<long long> coff; std::vector<uint32_t> pool;
	KswParams P = make_ksw_params(m, mat, q, e, w, zdrop, flag, 0);
	P.codes_ok = codes_below(queries, (size_t)q_off[n], m) && codes_below(targets, (size_t)t_off[n], m);
	rc = run_ksw_jobs(jobs, d_q.as<uint8_t>(), d_t.as<uint8_t>(), P, out, coff, pool, mat);
	if (rc) return rc;
	int64_t used = 0; int ret = 0;
	for (int i = 0; i < n; ++i) {
		const KswOut &o = out[i];
		ez[i].max = o.max; ez[i].zdropped = o.zdropped; ez[i].max_q = o.max_q; ez[i].max_t = o.max_t;
		ez[i].mqe = o.mqe; ez[i].mqe_t = o.mqe_t; ez[i].mte = o.mte; ez[i].mte_q = o.mte_q;
		ez[i].score = o.score; ez[i].n_cigar = o.n_cigar;
		if (o.n_cigar > 0) {
			if (used + o.n_cigar <= cigar_cap) memcpy(cigar + used, pool.data() + coff[i], sizeof(uint32_t) * (size_t)o.n_cigar);
			else ret = IHP_E_CAPACITY;
			used += o.n_cigar;
		}
		cigar_off[i + 1] = used;
	}
	return ret;
}

// Diagnostics: n (read, target 0, target 1) items through the alignment fallback's two-target sweep (ksw_duo.h) -- what
// k_fallback runs per (event, read), here on caller-made strings so that a test can compare it with the compiled reference.
// ez[2 i], ez[2 i + 1]: max / max_q / max_t / n_cigar of the item's alignments (the other fields stay reset: the fallback
// reads none of them); n_cigar = -2 marks an item the sweep does not take (ksw_duo_ok).  cigar: 2 n slots of cig_slot words.
extern "C" int ihp_debug_ksw_duo_batch(int32_t n, const uint8_t *reads, const int64_t *q_off, const uint8_t *t0, const int64_t *t0_off,
                                       const uint8_t *t1, const int64_t *t1_off, int8_t m, const int8_t *mat, int8_t q, int8_t e,
                                       int w, int zdrop, int flag, ihp_ez *ez, uint32_t *cigar, int32_t cig_slot)
{
	if (n < 0 || !q_off || !t0_off || !t1_off || !mat || !ez || !cigar || cig_slot < 1 || (n && (!reads || !t0 || !t1))) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	if (n == 0) return 0;
	const KswParams P = make_ksw_params(m, mat, q, e, w, zdrop, flag, 0);
	int qmax = 1, tmax = 1;
	for (int i = 0; i < n; ++i) {
		qmax = std::max(qmax, (int)(q_off[i + 1] - q_off[i]));
		tmax = std::max(tmax, (int)std::max(t0_off[i + 1] - t0_off[i], t1_off[i + 1] - t1_off[i]));
	}
	qmax = std::min(qmax, 64 * DUO_NS_MAX);
	const size_t lds = std::min(ksw_duo_lds_bytes(tmax), (size_t)g.max_lds - 2048), p_cap = ksw_duo_p_bytes(qmax, tmax) + 64;
	const int grid = std::min(n, 64), cig_cap = qmax + tmax + 16;
	DBuf d_q, d_t0, d_t1, d_qo, d_t0o, d_t1o, d_p, d_ct, d_ez, d_cig;
	if ((rc = d_q.upload(reads, (size_t)q_off[n], g.stream)) || (rc = d_t0.upload(t0, (size_t)t0_off[n], g.stream)) || (rc = d_t1.upload(t1, (size_t)t1_off[n], g.stream))) return rc;
	if ((rc = d_qo.upload(q_off, sizeof(int64_t) * (n + 1), g.stream)) || (rc = d_t0o.upload(t0_off, sizeof(int64_t) * (n + 1), g.stream)) ||
	    (rc = d_t1o.upload(t1_off, sizeof(int64_t) * (n + 1), g.stream))) return rc;
	if ((rc = d_p.alloc(p_cap * grid)) || (rc = d_ct.alloc(sizeof(uint32_t) * (size_t)cig_cap * grid)) || (rc = d_ez.alloc(sizeof(KswOut) * 2 * n)) ||
	    (rc = d_cig.alloc(sizeof(uint32_t) * (size_t)cig_slot * 2 * n))) return rc;
	DuoTestArgs a;
	a.n = n; a.q = d_q.as<uint8_t>(); a.t0 = d_t0.as<uint8_t>(); a.t1 = d_t1.as<uint8_t>();
	a.q_off = d_qo.as<long long>(); a.t0_off = d_t0o.as<long long>(); a.t1_off = d_t1o.as<long long>();
	a.P = P; a.lds_budget = (int)lds; a.p_scratch = d_p.as<uint8_t>(); a.p_cap = p_cap; a.cig_tmp = d_ct.as<uint32_t>(); a.cig_cap = cig_cap;
	a.ez = d_ez.as<KswOut>(); a.cig = d_cig.as<uint32_t>(); a.cig_slot = cig_slot;
	hipLaunchKernelGGL(k_ksw_duo_test, dim3(grid), dim3(64), lds + 64, g.stream, a);
	HIPC(hipGetLastError());
	std::vector<KswOut> out(2 * (size_t)n);
	HIPC(hipMemcpyAsync(out.data(), d_ez.p, sizeof(KswOut) * out.size(), hipMemcpyDeviceToHost, g.stream));
	HIPC(hipMemcpyAsync(cigar, d_cig.p, sizeof(uint32_t) * (size_t)cig_slot * 2 * n, hipMemcpyDeviceToHost, g.stream));
	HIPC(hipStreamSynchronize(g.stream));
	for (size_t i = 0; i < out.size(); ++i) {
		const KswOut &o = out[i];
		ez[i].max = o.max; ez[i].zdropped = o.zdropped; ez[i].max_q = o.max_q; ez[i].max_t = o.max_t;
		ez[i].mqe = o.mqe; ez[i].mqe_t = o.mqe_t; ez[i].mte = o.mte; ez[i].mte_q = o.mte_q; ez[i].score = o.score; ez[i].n_cigar = o.n_cigar;
	}
	return 0;
}

// Drop-in for the reference's FFI seam (ksw2.h:54).
extern "C" void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                              int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                              ksw_extz_t *ez)
{
	(void)km;
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;          // ksw_reset_extz, ksw2_extz2_sse.c:81-86
	ez->max = 0; ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0; ez->zdropped = 0;
	g_ksw_status = 0;
	if (m <= 0 || qlen <= 0 || tlen <= 0) return;
	int64_t qo[2] = {0, qlen}, to[2] = {0, tlen}, co[2];
	ihp_ez r;
	std::vector<uint32_t> cig((size_t)qlen + tlen + 4);
	const int rc = ihp_ksw_extz2_batch(1, query, qo, target, to, m, mat, q, e, w, zdrop, flag, &r, cig.data(), (int64_t)cig.size(), co);
	if (rc) {
		// The reference's signature has no return code and a reset ez reads as "no alignment": make the failure loud.
		// The code stays in ihp_ksw_last_status(), the text goes to stderr; IHP_KSW_STRICT=1 aborts like the reference's
		// own assert (ksw2_extz2_sse.c:237) would.
		g_ksw_status = rc;
		if (rc != IHP_E_HIP) snprintf(g.err, sizeof(g.err), "ksw_extz2_sse: %s (qlen %d, tlen %d, w %d, flag 0x%x)", ihp_strerror(rc), qlen, tlen, w, flag);
		fprintf(stderr, "indelope_hip: ksw_extz2_sse FAILED, ez left reset: %s\n", g.err);
		const char *strict = getenv("IHP_KSW_STRICT");
		if (g_knob.strict_ksw || (strict && strict[0] == '1')) abort();
		return;
	}
	ez->max = (uint32_t)r.max; ez->zdropped = (uint32_t)r.zdropped; ez->max_q = r.max_q; ez->max_t = r.max_t;
	ez->mqe = r.mqe; ez->mqe_t = r.mqe_t; ez->mte = r.mte; ez->mte_q = r.mte_q; ez->score = r.score;
	if (r.n_cigar > ez->m_cigar) {                               // grow like ksw_push_cigar (:34-37): powers of two from 4
		int mc = ez->m_cigar ? ez->m_cigar : 4;
		while (mc < r.n_cigar) mc <<= 1;
		ez->cigar = (uint32_t *)realloc(ez->cigar, (size_t)mc << 2);
		ez->m_cigar = mc;
	}
	if (r.n_cigar > 0) memcpy(ez->cigar, cig.data(), sizeof(uint32_t) * (size_t)r.n_cigar);
	ez->n_cigar = r.n_cigar;
}

extern "C" int ihp_ksw_last_status(void) { return g_ksw_status; }

// ------------------------------------------------------------ Contig API ops
static int contig_op(int op, ihp_contig *t, ihp_contig *q, int64_t min_overlap, int64_t max_mismatch, int rule,
                     const ihp_match *min_, ihp_match *mout, int64_t min_support)
{
	int rc = ensure_init();
	if (rc) return rc;
	const int64_t tl = t->len, ql = q ? q->len : 0;
	if (tl < 0 || ql < 0 || tl > MAXLEN || ql > MAXLEN) return IHP_E_CAPACITY;
	const int t_cap = ((int)tl + 3) / 4 * 4 + 16, q_cap = ((int)ql + 3) / 4 * 4 + 16;
	const int arena_cap = t_cap + q_cap + 3 * (int)(tl + ql) + 1024;
	const int corr_cap = (int)std::max<int64_t>(tl + ql, 16);
	DBuf d_seq, d_sup, d_corr, d_res;
	if ((rc = d_seq.alloc((size_t)arena_cap + 64))) return rc;
	if ((rc = d_sup.alloc(sizeof(uint32_t) * ((size_t)arena_cap + 64)))) return rc;
	if ((rc = d_corr.alloc(sizeof(Corr) * (size_t)corr_cap))) return rc;
	if ((rc = d_res.alloc(sizeof(long long) * 16))) return rc;
	if (tl) {
		HIPC(hipMemcpyAsync(d_seq.as<uint8_t>(), t->sequence, (size_t)tl, hipMemcpyHostToDevice, g.stream));
		HIPC(hipMemcpyAsync(d_sup.as<uint32_t>(), t->support, sizeof(uint32_t) * (size_t)tl, hipMemcpyHostToDevice, g.stream));
	}
	if (ql) {
		HIPC(hipMemcpyAsync(d_seq.as<uint8_t>() + t_cap, q->sequence, (size_t)ql, hipMemcpyHostToDevice, g.stream));
		HIPC(hipMemcpyAsync(d_sup.as<uint32_t>() + t_cap, q->support, sizeof(uint32_t) * (size_t)ql, hipMemcpyHostToDevice, g.stream));
	}
	OpArgs a;
	memset(&a, 0, sizeof(a));
	a.op = op; a.arena_seq = d_seq.as<uint8_t>(); a.arena_sup = d_sup.as<uint32_t>(); a.arena_cap = arena_cap;
	a.corr = d_corr.as<Corr>(); a.corr_cap = corr_cap;
	a.t_off = 0; a.t_len = (int)tl; a.t_cap = t_cap; a.t_nreads = t->nreads; a.t_start = t->start;
	a.q_off = t_cap; a.q_len = (int)ql; a.q_cap = q_cap; a.q_nreads = q ? q->nreads : 0; a.q_start = q ? q->start : 0;
	a.min_overlap = min_overlap; a.max_mismatch = max_mismatch; a.rule = rule; a.min_support = min_support;
	a.result = d_res.as<long long>();
	std::vector<Corr> hc;
	if (op == 1) {
		if (min_->offset == IHP_UNALIGNED) return 0;             // contig.nim:159
		const int64_t off = min_->offset, aoff = off < 0 ? -off : off;
		if (aoff > MAXLEN) return IHP_E_ARG;
		int64_t newlen = off < 0 ? std::max(aoff + tl, ql) : std::max(tl, off + ql);
		if (newlen > t->cap) return IHP_E_CAPACITY;
		if (min_->n_corrections > corr_cap) return IHP_E_CAPACITY;
		hc.resize((size_t)min_->n_corrections);
		for (int64_t i = 0; i < min_->n_corrections; ++i) {
			const ihp_correction &c = min_->corrections[i];
			if (c.qoff < 0 || c.qoff >= ql || c.toff < 0 || c.toff >= tl) return IHP_E_ARG;
			hc[i].qoff = (int)c.qoff; hc[i].toff = (int)c.toff; hc[i].qbest = c.qbest;
		}
		if (!hc.empty()) HIPC(hipMemcpyAsync(d_corr.p, hc.data(), sizeof(Corr) * hc.size(), hipMemcpyHostToDevice, g.stream));
		a.off = (int)off; a.ncorr = (int)hc.size();
	}
	hipLaunchKernelGGL(k_contig_op, dim3(1), dim3(64), 0, g.stream, a);
	HIPC(hipGetLastError());
	long long R[16];
	HIPC(hipMemcpyAsync(R, d_res.p, sizeof(R), hipMemcpyDeviceToHost, g.stream));
	HIPC(hipStreamSynchronize(g.stream));
	if (R[0]) return (int)R[0];
	if (op == 0) {
		mout->matches = R[2]; mout->mismatches = R[3]; mout->contig_i = -1;
		mout->offset = R[1] ? R[4] : IHP_UNALIGNED;
		mout->n_corrections = R[5];
		if (R[5] > mout->corr_cap) return IHP_E_CAPACITY;
		if (R[5]) {
			hc.resize((size_t)R[5]);
			HIPC(hipMemcpy(hc.data(), d_corr.p, sizeof(Corr) * hc.size(), hipMemcpyDeviceToHost));
			for (size_t i = 0; i < hc.size(); ++i) {
				mout->corrections[i].qoff = hc[i].qoff; mout->corrections[i].toff = hc[i].toff;
				mout->corrections[i].qbest = hc[i].qbest; mout->corrections[i]._pad = 0;
			}
		}
		return 0;
	}
	// insert / trim: copy the (possibly relocated) contigs back
	const int64_t ntl = R[7];
	if (ntl > t->cap) return IHP_E_CAPACITY;
	if (ntl) {
		HIPC(hipMemcpy(t->sequence, d_seq.as<uint8_t>() + R[6], (size_t)ntl, hipMemcpyDeviceToHost));
		HIPC(hipMemcpy(t->support, d_sup.as<uint32_t>() + R[6], sizeof(uint32_t) * (size_t)ntl, hipMemcpyDeviceToHost));
	}
	t->len = ntl; t->nreads = R[8]; t->start = R[9];
	if (op == 1 && ql) {
		HIPC(hipMemcpy(q->sequence, d_seq.as<uint8_t>() + R[10], (size_t)ql, hipMemcpyDeviceToHost));
		HIPC(hipMemcpy(q->support, d_sup.as<uint32_t>() + R[10], sizeof(uint32_t) * (size_t)ql, hipMemcpyDeviceToHost));
	}
	return 0;
}

extern "C" int ihp_slide_align(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                               int64_t max_mismatch, int allow_rule, ihp_match *out)
{
	if (!q || !t || !out) return IHP_E_ARG;
	if (min_overlap < -MAXLEN || min_overlap > (1 << 30) || max_mismatch < 0 || max_mismatch > (1 << 30)) return IHP_E_ARG;
	return contig_op(0, (ihp_contig *)t, (ihp_contig *)q, min_overlap, max_mismatch, allow_rule, nullptr, out, 0);
}

extern "C" int ihp_contig_insert(ihp_contig *t, ihp_contig *q, const ihp_match *m)
{
	if (!t || !q || !m) return IHP_E_ARG;
	return contig_op(1, t, q, 0, 0, 0, m, nullptr, 0);
}

extern "C" int ihp_contig_trim(ihp_contig *c, int64_t min_support)
{
	if (!c) return IHP_E_ARG;
	return contig_op(2, c, nullptr, 0, 0, 0, nullptr, nullptr, min_support);
}

// -------------------------------------------------------------- k-mer tally
static bool host_mincode(const char *kmer, int K, unsigned long long &code)
{
	unsigned long long f = 0, rc = 0;
	for (int i = 0; i < K; ++i) {
		int b;
		switch (kmer[i]) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; default: return false; }
		f = (f << 2) | (unsigned long long)b;
		rc |= (unsigned long long)(3 - b) << (2 * i);
	}
	code = f < rc ? f : rc;
	return true;
}

extern "C" int ihp_kmer_tally(int32_t n_reads, const uint8_t *bases, const int64_t *read_off,
                              const uint8_t *mapq, int32_t min_mapq, int32_t K,
                              const char *ref_kmer, const char *alt_kmer, int32_t counts[3])
{
	if (n_reads < 0 || !read_off || !ref_kmer || !alt_kmer || !counts || K < 1 || K > 31) return IHP_E_ARG;
	if ((int)strnlen(ref_kmer, (size_t)K) < K || (int)strnlen(alt_kmer, (size_t)K) < K) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	TallyOneArgs a;
	if (!host_mincode(ref_kmer, K, a.refe) || !host_mincode(alt_kmer, K, a.alte)) return IHP_E_UNSUPPORTED;
	DBuf d_b, d_o, d_m, d_c;
	if ((rc = d_b.upload(bases, (size_t)read_off[n_reads], g.stream))) return rc;
	if ((rc = d_o.upload(read_off, sizeof(int64_t) * (size_t)(n_reads + 1), g.stream))) return rc;
	if (mapq && (rc = d_m.upload(mapq, (size_t)n_reads, g.stream))) return rc;
	if ((rc = d_c.alloc(16))) return rc;
	a.bases = d_b.as<uint8_t>(); a.read_off = d_o.as<long long>(); a.mapq = mapq ? d_m.as<uint8_t>() : nullptr;
	a.n_reads = n_reads; a.min_mapq = min_mapq; a.K = K; a.counts = d_c.as<int>();
	hipLaunchKernelGGL(k_tally_one, dim3(1), dim3(64), 0, g.stream, a);
	HIPC(hipGetLastError());
	HIPC(hipMemcpyAsync(counts, d_c.p, 12, hipMemcpyDeviceToHost, g.stream));
	HIPC(hipStreamSynchronize(g.stream));
	return 0;
}

// ------------------------------------------------------- the batched region path
enum { WQ_SETS = 26 };      // work-queue counter sets: 11 assembly launches, ksw2, tally, fallback; [14] holds the combine cost-class counters; [15] the read-rich k_asm_reads launch; [16] the third combine tier; [17] the roomy ksw2 launch; [18] the pair launch of ksw2; [19..23] the zero block of the ksw2 plan (jobs per contig length, pairs, singles); [24] the wide combine launch (regions of more than 255 reads); [25] the roomy fallback launch
enum { M_CIG = 0, M_EV = 2, M_NJOBS = 4, M_CNT_ASM = 5, M_KSW_OVF = 6, M_CNT_TALLY = 7, M_OVF = 8, M_NRETRY = 11, M_CNT_RETRY = 12, M_NRETRY2 = 13, M_CNT_ASM2 = 14, M_NRETRY3 = 15, M_CNT_ASM3 = 16, M_NFB = 17, M_OVF_HIT = 18, M_NRETRY0 = 19, M_HIT = 20, M_NRETRYC = 22, M_NTIERB = 23, M_NTIERC = 24, M_NRECS = 25 /* 2 */, M_FB_OVF = 27, M_WORDS = 32,
       M_SLAB_BAD = 48, M_HIST = 49, M_MANYC = 60 };   // M_MANYC: regions with more contigs than the first tier's short table holds   // (behind the stamps, inside the report block: raised by k_slab_expand when a compact slab's lengths do not add up)
struct ihp_batch {
	ihp_params P;
	int R = 0; long long n_reads = 0, n_bases = 0, n_ref = 0;
	int max_region_bases = 0, max_read_len = 0, max_ref_len = 0;
	std::vector<int64_t> h_region_read_off, h_ref_origin;
	// inputs
	DBuf region_read_off, read_off, bases, quals, read_start, read_stop, mapq, read_skip, ref_off, ref_bases, ref_origin;
	DBuf trim_lo, trim_hi;
	DBuf in_slab, bases4;                                  // ihp_batch_upload_slab: every input in one device buffer; BAM 4-bit bases (k_prepack writes the ASCII ones)
	bool has_quals = false, has_skip = false, has_trim = false, has_b4 = false;
	bool has_pk2 = false;                                  // the compact slab brought the reads 2 bits each in the packed form itself (IHP_SLAB2_BASES_2BIT): v2_pk is a view of the slab
	bool slab_bad = false;                                 // a compact slab whose lengths did not add up (k_slab_expand): every wait reports IHP_E_ARG
	int fetch_flags = 0;                                   // IHP_FETCH_*
	// scratch
	DBuf arena_seq, arena_sup, lds_sup, lds_sup2, corr, p_scratch, cig_tmp, misc, prof, retry_list, retry_list2;
	int grid_retry = 0, grid_asm2 = 0, grid_asm3 = 0, lds_arena1 = 0, lds_arena2 = 0, lds_arena3 = 0;
	DBuf lds_sup3, retry_list3, corr2, cls_list, cls_n, aux;
	void *aux_host = nullptr;                              // page-locked staging block of `aux` (from the slab cache)
	// packed read phase (asm2_dev.h): per-read outputs of k_prepack (they persist with the inputs) and the pass's sizes
	DBuf v2_pk, v2_trim_lo, v2_trim_hi, v2_read_bad, retry_list0, v2_hoff, v2_hand, lpt_seg;
	bool v2 = false; int v2_arena = 0, v2_pdw = 0, v2_pm = 0, v2_arena_big = 0, v2_pm_big = 0, v2_arena_b = 0, v2_pm_b = 0, v2_arena_c = 0, v2_pm_c = 0, grid_v2c = 0, grid_v2 = 0, grid_v2b = 0, grid_v2r = 0, grid_v2big = 0, grid_pack = 0, grid_ovf1 = 0;
	DBuf retry_listc;
	long long v2_hand_dwords = 0;
	int n_cls[4] = {0, 0, 0, 0};                           // regions per assembly class (host prediction from the read bases)
	int n_small = 0, n_rich = 0;                           // class 1 = the regions of the usual size + the read-rich ones the packed path takes (its own k_asm_reads launch)
	int n_deep = 0;                                        // regions of the packed path with more than 256 reads: their combine runs the wide build (16-bit supports)
	int v2_arena_deep = 0, v2_pm_deep = 0, grid_v2deep = 0;
	int v2_pdw_rich = 0, grid_v2r_rich = 0;
	hipEvent_t ev_rfork = nullptr, ev_rjoin = nullptr;
	hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_bfork = nullptr, ev_bjoin = nullptr, ev_kfork = nullptr, ev_kjoin = nullptr;
	int grid_asm = 0, grid_ksw = 0, grid_tally = 0;
	int arena_cap = 0, stage_cap = 0, corr_cap = 0, lds_ksw = 0, cig_cap = 0;
	size_t p_cap = 0;
	// two alignments per wavefront (ksw_pair.h): the plan's lists, the pair launch's LDS per wave and scratch per workgroup
	DBuf ksw_plan, p_scratch_pair, cig_tmp_pair;
	int lds_ksw_pair = 0; size_t p_cap_pair = 0;           // 0: no pair launch for this batch
	unsigned long long hint_key = 0;                       // the batch's shape (HintTable)
	long long cig_pool_cap = 0, cig_bump_cap = 0, ev_pool_cap = 0, njobs_cap = 0;
	// alignment fallback (indelope.nim:312-372)
	DBuf fb_items, fb_p_scratch, fb_cig_tmp, fb_ovf;       // fb_ovf: items the main fallback launch could not hold (its roomy launch shares the roomy ksw2 launch's scratch)
	DBuf pack_cnt, pack_slab;                              // result compaction (ihp_batch_fetch)
	DBuf tally_recs, tally_ovf; int tally_rec_cap = 0;     // k_tally_prep -> k_tally (TallyRec)
	DBuf hit_pool; long long hit_cap = 0;                  // first-hit k-mer positions per (tallied event, read)
	bool timing = false;                                   // device wall-clock stamps: start / end of the four stages
	// everything a run expects to be zero lives in ONE buffer (`misc`): [counters | stamps | work queues | per-region hit
	// counts].  It is cleared once at upload; after that the last kernel of every run (k_summary) copies the first
	// REPORT_INTS ints to `report` and clears the buffer for the next run -- a run has no memset.
	static constexpr size_t Z_TIMES = 128, Z_QUEUES = 256;
	size_t z_hitcnt() const { return Z_QUEUES + sizeof(int) * WQ_WORDS * WQ_SETS; }
	size_t z_bytes() const { return z_hitcnt() + sizeof(int) * (size_t)std::max(R, 1); }
	unsigned long long *times_dev() const { return (unsigned long long *)((char *)misc.p + Z_TIMES); }
	int *queues_dev() const { return (int *)((char *)misc.p + Z_QUEUES); }
	int *hitcnt_dev() const { return (int *)((char *)misc.p + z_hitcnt()); }
	int grid_fb = 0, lds_fb = 0, fb_cig_cap = 0, max_region_reads = 0;
	size_t fb_p_cap = 0, fb_p_cap_big = 0; int fb_cig_cap_big = 0;
	static constexpr int FB_OVF_CAP = 1 << 16;
	// outputs
	DBuf status, n_pre, n_final, ctg_start, ctg_nreads, ctg_seq_off, ctg_len, aln_flags, aln_ref_len, aln_ref_start;
	DBuf out_seq, out_sup, jobs, ez, cig_off, cig_pool, ev_off, n_ev, ev_pool, summary;
	hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	hipStream_t stream = nullptr;
	bool ran = false, work_live = false;
	double acc_ms[4] = {0, 0, 0, 0}; long long acc_n = 0;  // stage times summed over the runs since the last reset (ihp_batch_kernel_ms_mean)
	bool acc_pending = false;                              // the last run's stamps have not been added yet
	bool spec_skipped = false;                             // the run left out the retry launches (nobody needed them in the last batch): checked when it is waited for
	bool force_full = false;
	bool ksw_skipped = false;                              // the run left out the roomy ksw2 launch (no job needed it in the last batch): checked at the wait
	bool fb_roomy_skipped = false;                         // the same for the roomy launch of the alignment fallback
	DBuf ksw_ovf, p_scratch_big, cig_tmp_big;              // jobs the main ksw2 launch could not hold, and the roomy launch's scratch
	size_t p_cap_big = 0; int cig_cap_big = 0, grid_kovf = 0;
	bool tier_wide = false;                                // the first combine tier runs the build with the full contig table (many regions with more than COMB_MAXC_A contigs)
	long long v2_nb1 = 0; int tier_occ = 0, tier_occ_default = 0, tier_sig = 0;   // first-tier sizing of the combine launches (size_combine_tiers)
	bool counted = false;                                  // k_pack_count / k_pack_scan of the last run are enqueued (or done)
	mutable bool hint_counted = false;                     // the last run has been added to the shape's streaks (TierHint::clean_*): once per run, however many waits confirm it
	long long n_reruns = 0;
	bool dirty = false;                                    // a run was cut short after some launches: `misc` is not known to be clear
	int *report = nullptr;                                 // page-locked host block: the last run's counters, flags and stamps
	int grid_ovf2 = 0, grid_ovf3 = 0, grid_ovf4 = 0;       // grids of the run-time overflow launches
	ihp_batch() {
		// the second batch alive beside another: their launch chains only overlap reliably with more hardware queues than the
		// runtime's default of four (see ihp_init) -- said once, unconditionally, because the loss (-15 %) is otherwise silent
		static std::atomic<bool> told{false};
		if (g_live_batches.fetch_add(1) >= 1 && !getenv("GPU_MAX_HW_QUEUES") && !told.exchange(true))
			fprintf(stderr, "[ihp] note: GPU_MAX_HW_QUEUES is not set and more than one batch is alive: with the runtime's default of 4 hardware queues "
			                "the launch chains of different batches can share a queue and run one after the other; export GPU_MAX_HW_QUEUES=16 before the "
			                "process's first HIP call (include/indelope_hip.h)\n");
	}
	~ihp_batch() {
		// the buffers go back to the pool (the members are released after this body): nothing of this batch may still be running
		if (stream2) { (void)hipStreamSynchronize(stream2); g_streams.put(stream2); }
		if (stream) { (void)hipStreamSynchronize(stream); g_streams.put(stream); }
		// only now can the report slot be handed to another batch: a run in flight would still write its counters there
		if (report) g_reports.put(report);
		if (aux_host) g_slabs.put(aux_host);
		g_live_batches.fetch_sub(1);
		if (ev_fork) (void)hipEventDestroy(ev_fork);
		if (ev_join) (void)hipEventDestroy(ev_join);
		if (ev_bfork) (void)hipEventDestroy(ev_bfork);
		if (ev_bjoin) (void)hipEventDestroy(ev_bjoin);
		if (ev_kfork) (void)hipEventDestroy(ev_kfork);
		if (ev_kjoin) (void)hipEventDestroy(ev_kjoin);
		if (ev_rfork) (void)hipEventDestroy(ev_rfork);
		if (ev_rjoin) (void)hipEventDestroy(ev_rjoin);
		for (auto &e : ev) if (e) (void)hipEventDestroy(e);
	}
};

// misc layout (ints): [0..1] cigar cursor (u64), [2..3] event cursor (u64), [4] n_jobs,
// [5] asm counter, [6] ksw counter, [7] tally counter, [8..10] overflow flags

// Scratch and result buffers of a batch (everything but its inputs and the small per-run state): taken from the
// caching pool at upload, handed back by ihp_batch_release_outputs and taken again by the next ihp_batch_run.
static int alloc_work(ihp_batch *b)
{
	if (b->work_live) return 0;
	int rc;
	const ihp_params *p = &b->P;
	const int R = b->R;
	const long long slots = b->n_reads, NR = b->n_reads;
#define AL(buf, bytes) do { if ((rc = b->buf.alloc((size_t)(bytes)))) return rc; } while (0)
	AL(arena_seq, (size_t)b->arena_cap * b->grid_retry);
	AL(arena_sup, sizeof(uint32_t) * (size_t)b->arena_cap * b->grid_retry);
	AL(lds_sup, sizeof(uint32_t) * (size_t)b->lds_arena1 * b->grid_asm);
	AL(lds_sup2, sizeof(uint32_t) * (size_t)b->lds_arena2 * b->grid_asm2);
	AL(lds_sup3, sizeof(uint32_t) * (size_t)b->lds_arena3 * b->grid_asm3);
	AL(retry_list, sizeof(int) * (size_t)R);
	if (b->v2) { AL(retry_list0, sizeof(int) * (size_t)R); AL(v2_hand, sizeof(uint32_t) * (size_t)b->v2_hand_dwords);
		AL(retry_listc, sizeof(int) * (size_t)R); AL(lpt_seg, sizeof(int) * (size_t)R * LPT_CLASSES * LPT_TIERS); }
	AL(retry_list2, sizeof(int) * (size_t)R);
	AL(retry_list3, sizeof(int) * (size_t)R);
	AL(corr2, sizeof(Corr) * (size_t)b->corr_cap * std::max(std::max(b->grid_asm2, b->grid_asm3), std::max(b->grid_retry, b->grid_v2b)));
	AL(corr, sizeof(Corr) * (size_t)b->corr_cap * std::max(std::max(std::max(b->grid_asm, std::max(std::max(b->grid_v2, b->grid_v2b), b->grid_v2big)), b->grid_asm2), std::max(b->grid_asm3, b->grid_retry)));
	AL(p_scratch, b->p_cap * b->grid_ksw);
	AL(cig_tmp, sizeof(uint32_t) * (size_t)b->cig_cap * b->grid_ksw);
	AL(ksw_ovf, sizeof(int) * (size_t)std::max<long long>(1, b->njobs_cap));
	// the tally's records: the jobs with events -- a few per region; what does not fit takes k_tally's own header path
	b->tally_rec_cap = (int)std::min<long long>(std::max<long long>(1, b->njobs_cap), g_knob.tally_rec_cap > 0 ? g_knob.tally_rec_cap : 4ll * b->R + 4096);
	AL(tally_recs, sizeof(TallyRec) * (size_t)b->tally_rec_cap);
	AL(tally_ovf, sizeof(int) * (size_t)std::max<long long>(1, b->njobs_cap));
	if (b->p_cap_pair) {
		AL(ksw_plan, sizeof(int) * ksw_plan_ints(std::max<long long>(1, b->njobs_cap)));
		AL(p_scratch_pair, b->p_cap_pair * b->grid_ksw);
		AL(cig_tmp_pair, sizeof(uint32_t) * (size_t)b->cig_cap * b->grid_ksw);
	}
	// (the roomy ksw2 launch's scratch -- up to 512 MB -- is taken when that launch is enqueued: most runs leave it out)
	if (p->fallback) {
		AL(fb_items, sizeof(FbItem) * (size_t)b->ev_pool_cap);
		AL(fb_p_scratch, b->fb_p_cap * b->grid_fb);
		AL(fb_cig_tmp, sizeof(uint32_t) * (size_t)b->fb_cig_cap * b->grid_fb);
		AL(fb_ovf, sizeof(int) * (size_t)ihp_batch::FB_OVF_CAP);
	}
	AL(prof, sizeof(long long) * 64);
	AL(status, sizeof(int) * R); AL(n_pre, sizeof(int) * R); AL(n_final, sizeof(int) * R);
	AL(ctg_start, 8 * slots); AL(ctg_nreads, 8 * slots); AL(ctg_seq_off, 8 * slots);
	AL(ctg_len, 4 * slots); AL(aln_flags, 4 * slots); AL(aln_ref_len, 4 * slots); AL(aln_ref_start, 8 * slots);
	AL(out_seq, b->n_bases); AL(out_sup, 4 * (size_t)b->n_bases);
	AL(jobs, sizeof(AlnJob) * slots); AL(ez, sizeof(KswOut) * slots); AL(cig_off, 8 * slots);
	AL(cig_pool, 4 * (size_t)b->cig_pool_cap);
	AL(ev_off, 8 * slots); AL(n_ev, 4 * slots); AL(ev_pool, sizeof(DevEvent) * (size_t)b->ev_pool_cap);

	// HIT_SLOTS events per region at fixed places (2 x nreads ints each, region r at 8 x its first read index), then
	// a bump region of the same size for regions with more tallied events (hit_cap, set at upload)
	AL(hit_pool, sizeof(int) * (size_t)b->hit_cap);
#undef AL
	(void)NR; (void)slots;
	b->work_live = true;
	return 0;
}

static void release_work(ihp_batch *b)
{
	DBuf *bufs[] = {&b->retry_list0, &b->retry_listc, &b->lpt_seg, &b->v2_hand, &b->arena_seq, &b->arena_sup, &b->lds_sup, &b->lds_sup2, &b->lds_sup3, &b->retry_list, &b->retry_list2, &b->retry_list3,
	                &b->corr2, &b->corr, &b->p_scratch, &b->cig_tmp, &b->ksw_ovf, &b->ksw_plan, &b->p_scratch_pair, &b->cig_tmp_pair, &b->p_scratch_big, &b->cig_tmp_big, &b->fb_items, &b->fb_p_scratch, &b->fb_cig_tmp, &b->fb_ovf, &b->prof,
	                &b->status, &b->n_pre, &b->n_final, &b->ctg_start, &b->ctg_nreads, &b->ctg_seq_off, &b->ctg_len, &b->aln_flags,
	                &b->aln_ref_len, &b->aln_ref_start, &b->out_seq, &b->out_sup, &b->jobs, &b->ez, &b->cig_off, &b->cig_pool,
	                &b->ev_off, &b->n_ev, &b->ev_pool, &b->hit_pool, &b->pack_cnt, &b->pack_slab, &b->tally_recs, &b->tally_ovf};
	for (DBuf *d : bufs) d->release();
	b->work_live = false;
}

// LDS sizing of the combine launches (k_asm_combine3, asm3_dev.h).  Capacity C = C bytes of supports (kept for multi-read
// contigs only) + C / 8 + 128 dwords of packed bases (every contig) beside the kernel's static LDS (asked of the runtime: a
// stale constant here once cost every tier a wave per CU).
// The hardware hands LDS to a workgroup in granules of 1280 bytes on gfx950 (160 KB = 128 granules): a wave whose static +
// dynamic LDS is one byte over k granules costs k + 1, so only the occupancies 128 / k exist -- 25, 21, 18, 16, 14, 12, 11, 10,
// 9, 8 ... per CU.  (Measured on the first tier, C2: 7652 bytes per wave -> 21 resident, 2.70 ms per 100 000 regions; 7748
// bytes -> 18 resident, 2.89 ms.  Round 4 sized the arenas as if any number of waves could share the 160 KB: its "17 per CU"
// were 16, and a tier cut "for 24" ran 21 with an arena a granule short of what 21 could have had.)
constexpr long long LDS_GRAN = 1280;
static long long comb_stat() { return g.comb_static + 16; }                                 // static LDS of the full-table builds (+ alignment of the dynamic part)
static long long comb_stat_a() { return std::max(g.comb_static_a, 512) + 16; }               // ... of the first tier's short-table build (COMB_MAXC_A contigs)
static long long comb_pm_of(long long C) { return C / 8 + 128; }
static long long comb_wave_bytes(long long C, long long stat) { return (stat + C + 4 * comb_pm_of(C) + LDS_GRAN - 1) / LDS_GRAN * LDS_GRAN; }   // what the hardware sets aside
static int comb_occ_of(long long C, long long stat) { return (int)std::max<long long>(1, (g.max_lds / LDS_GRAN) / (comb_wave_bytes(C, stat) / LDS_GRAN)); }
// the largest capacity with which `occ` waves share a CU
static long long comb_cap_for(int occ, long long stat = -1)
{
	if (stat < 0) stat = comb_stat();
	const long long budget = std::max<long long>(1, (g.max_lds / LDS_GRAN) / std::max(1, occ)) * LDS_GRAN;
	return std::max<long long>(256, (budget - stat - 4 * 128) * 2 / 3 / 16 * 16 - 16);
}
static_assert(M_HIST + HIST_N <= M_MANYC && M_MANYC < REPORT_INTS, "the tier histogram leaves the report block");

// The three tiers and the roomy launch for a first tier of at most `occ_first` waves per CU (0: what the read bases of the
// usual region suggest).  Called at upload, and again by ihp_batch_run when the last batch of this shape showed that most
// regions need more than the first tier holds (see TierHint): C5's 300 bp reads leave 85 % of the regions above the arena
// that 0.3 x read bases predicts, and one launch at 8 waves per CU for all of them beats 10 waves for a few + 7 for the rest.
static void size_combine_tiers(ihp_batch *b, int occ_first)
{
	const int R = b->R;
	const long long nb1 = b->v2_nb1, stat = comb_stat(), stat_a = b->tier_wide ? comb_stat() : comb_stat_a();
	// Regions per CU.  The launch is bound by the latency of every region's chain and runs in proportion to the regions a CU
	// holds until about 20 of them (C2, per 100 000 regions: 10 per CU 4.53 ms, 12: 3.87, 14: 3.43, 16: 3.11, 18: 2.89, 20: 2.61;
	// a build for six waves per SIMD -- 80 VGPRs, 19 of them spilled -- with 21: 2.64, with 24: 2.73;
	// profiles/r05_combine_occupancy.txt): round 5 gave the first tier a build with a contig table of 32 entries (1.7 KB instead
	// of 3.2; regions with more contigs are filed under the second tier, or the first tier runs the full-table build when they
	// are many) and sizes every tier in whole LDS granules: six a wave, 20 regions per CU at the build's 95 VGPRs.
	// (Round 3 took 14 for launches of about one round of regions per wave slot: such a launch lasts as long
	// as its heaviest regions, and a caller that waited for every batch before starting the next saw those run faster with
	// fewer waves beside them.  A caller that keeps batches in flight -- a sweep, bench.py since round 4 -- has another
	// chain's kernels in those tails.)
	const int occ_hw = 4 * std::max(5, std::min(7, g_knob.comb_minw));
	const int occ_max = g_knob.comb_occ ? g_knob.comb_occ : 21;   // (six LDS granules a wave; the five-wave build keeps 20 of them resident)
	long long need_C = std::max<long long>(1024, (nb1 * 30 / 100 + 512 + 15) / 16 * 16);   // the usual region needs 0.2-0.3 of its read bases in these units; the rest goes to the roomier launches
	int occ_c = std::max(1, std::min(occ_max, comb_occ_of(need_C, stat_a)));
	// occ_first: what the last batch of this shape needed (the tier histogram) -- below OR above what the read bases suggest
	if (occ_first > 0) { occ_c = std::max(1, std::min(occ_max, occ_first)); need_C = 1024; }
	need_C = std::max(need_C, comb_cap_for(occ_c, stat_a));
	b->tier_occ = occ_c;
	// the roomy launch for regions whose contigs do not fit the first one's arena (many single-read contigs)
	b->v2_arena_big = (int)std::min<long long>(comb_cap_for(2), std::max<long long>(4 * need_C, (nb1 + 1024 + 15) / 16 * 16));
	b->grid_v2big = grid_for(R, std::max(1, std::min<int>(2, comb_occ_of(b->v2_arena_big, stat))));
	// the second and third tier: regions whose contigs (known when the read phase ends) need more than the first arena
	// -- many single-read contigs, long reads -- at about two thirds and a third of its occupancy
	{
		const int occ_b = std::max(1, (occ_c * 2 + 1) / 3), occ_t = std::max(1, occ_c / 3);
		b->v2_arena_b = (int)std::max(need_C, comb_cap_for(occ_b));
		b->grid_v2b = std::min(grid_for(R, occ_b), std::max(1, b->n_cls[0]));
		b->v2_arena_c = (int)std::max<long long>(b->v2_arena_b, comb_cap_for(occ_t));
		b->grid_v2c = std::min(grid_for(R, occ_t), std::max(1, b->n_cls[0]));
	}
	b->v2_arena = g_knob.v2_arena ? g_knob.v2_arena / 16 * 16 : (int)need_C;
	// the wide build's launch (regions of more than 255 reads): 2 bytes per support.  A deep pile-up holds about the contigs of a
	// usual one -- the reads all cover the same few hundred bases -- plus a single-read contig per read with an error, so the
	// arena of the usual region's second tier, at least, at eight regions per CU (their chains are the longest of the batch).
	if (b->n_deep) {
		const long long statw = g.comb_static_w + 16;
		const int occ_d = 8;
		const long long budget = std::max<long long>(1, (g.max_lds / LDS_GRAN) / occ_d) * LDS_GRAN;
		long long Cd = std::max<long long>(1024, (budget - statw - 4 * 128) * 2 / 5 / 16 * 16 - 16);      // 2 C + 4 (C / 8 + 128) bytes
		Cd = std::max<long long>(Cd, b->v2_arena_b);
		const int dyn_max_d = g.max_lds - 8192 - 1024;
		while (Cd > 1024 && 2 * Cd + 4 * comb_pm_of(Cd) > dyn_max_d) Cd -= 256;
		b->v2_arena_deep = (int)Cd; b->v2_pm_deep = (int)comb_pm_of(Cd);
		const int occ_real = (int)std::max<long long>(1, (g.max_lds / LDS_GRAN) / ((statw + 2 * Cd + 4 * comb_pm_of(Cd) + LDS_GRAN - 1) / LDS_GRAN));
		b->grid_v2deep = std::min(grid_for(b->n_deep, std::min(occ_real, 16)), std::max(1, b->n_deep));
	}
	// every combine launch must fit what hipFuncSetAttribute allows (max_lds - 8192 of dynamic LDS beside the static part):
	// the later tiers and the roomy launch give up arena first, the packed path is switched off only when the first tier does not fit
	const int dyn_max = g.max_lds - 8192 - 1024;
	while (b->v2_arena_b > 1024 && b->v2_arena_b + 4 * comb_pm_of(b->v2_arena_b) > dyn_max) b->v2_arena_b -= 256;
	while (b->v2_arena_c > 1024 && b->v2_arena_c + 4 * comb_pm_of(b->v2_arena_c) > dyn_max) b->v2_arena_c -= 256;
	b->v2_pm_c = (int)comb_pm_of(b->v2_arena_c);
	while (b->v2_arena_big > 1024 && b->v2_arena_big + 4 * comb_pm_of(b->v2_arena_big) > dyn_max) b->v2_arena_big -= 256;
	b->v2_pm = (int)comb_pm_of(b->v2_arena); b->v2_pm_b = (int)comb_pm_of(b->v2_arena_b); b->v2_pm_big = (int)comb_pm_of(b->v2_arena_big);
	b->grid_v2 = std::min(grid_for(R, std::max(1, std::min(g_knob.asm_waves ? g_knob.asm_waves : occ_hw, comb_occ_of(b->v2_arena, stat_a)))), std::max(1, b->n_cls[0]));
}

// The compact slab's host view (ihp_batch_upload_slab2): what batch_upload_common reads on the host comes from here when set.
struct Slab2In {
	const void *slab; const ihp_slab2_layout *L; int flags;
	const int64_t *region_base_off; const uint16_t *len;
};
static int batch_upload_common(const ihp_params *p, const ihp_batch_in *in, const void *slab, const ihp_slab_layout *SL, ihp_batch **bout, const Slab2In *s2 = nullptr);

extern "C" int ihp_batch_upload(const ihp_params *p, const ihp_batch_in *in, ihp_batch **bout)
{
	if (!in || (in->n_reads > 0 && !in->bases)) return IHP_E_ARG;
	return batch_upload_common(p, in, nullptr, nullptr, bout);
}

// Section offsets (bytes, 64-byte aligned) of the one-slab batch input; see include/indelope_hip.h.
extern "C" int ihp_slab_layout_for(int32_t n_regions, int64_t n_reads, int64_t n_bases, int64_t n_ref, ihp_slab_layout *L)
{
	if (!L || n_regions < 0 || n_reads < 0 || n_bases < 0 || n_ref < 0) return IHP_E_ARG;
	int64_t o = 0;
	auto sec = [&](int64_t bytes) { const int64_t at = o; o += (bytes + 63) / 64 * 64; return at; };
	L->region_read_off = sec(8 * ((int64_t)n_regions + 1)); L->read_off = sec(8 * (n_reads + 1));
	L->read_start = sec(8 * n_reads); L->read_stop = sec(8 * n_reads);
	L->ref_off = sec(8 * ((int64_t)n_regions + 1)); L->ref_origin = sec(8 * (int64_t)n_regions);
	L->trim_lo = sec(4 * n_reads); L->trim_hi = sec(4 * n_reads);
	L->mapq = sec(n_reads); L->read_skip = sec(n_reads);
	L->ref_bases = sec(n_ref + 64);
	L->bases4 = sec((n_bases >> 1) + n_reads + 64);
	L->bytes = o;
	return 0;
}

// Section offsets of the compact slab (include/indelope_hip.h, ihp_slab2_layout).
extern "C" int ihp_slab2_layout_for(int32_t n_regions, int64_t n_reads, int64_t n_bases, int64_t n_ref, int32_t flags, ihp_slab2_layout *L)
{
	if (!L || n_regions < 0 || n_reads < 0 || n_bases < 0 || n_ref < 0 || (flags & ~(IHP_SLAB2_REF_2BIT | IHP_SLAB2_BASES_2BIT))) return IHP_E_ARG;
	int64_t o = 0;
	auto sec = [&](int64_t bytes) { const int64_t at = o; o += (bytes + 63) / 64 * 64; return at; };
	L->region_read_off = sec(8 * ((int64_t)n_regions + 1)); L->region_base_off = sec(8 * ((int64_t)n_regions + 1));
	L->ref_off = sec(8 * ((int64_t)n_regions + 1)); L->ref_origin = sec(8 * (int64_t)n_regions);
	L->start_rel = sec(4 * n_reads);
	L->len = sec(2 * n_reads); L->span = sec(2 * n_reads); L->trim_lo = sec(2 * n_reads); L->trim_hi = sec(2 * n_reads);
	L->mapq = sec(n_reads); L->rflags = sec(n_reads);
	L->ref_packed = sec(((flags & IHP_SLAB2_REF_2BIT) ? (n_ref >> 2) : (n_ref >> 1)) + n_regions + 64);
	L->bases4 = sec((flags & IHP_SLAB2_BASES_2BIT) ? 4 * ((n_bases >> 4) + n_reads + 4) + 64 : (n_bases >> 1) + n_reads + 64);
	L->bytes = o;
	return 0;
}

extern "C" int ihp_batch_upload_slab2(const ihp_params *p, int32_t n_regions, int64_t n_reads, const void *slab, const ihp_slab2_layout *L,
                                      int32_t flags, ihp_batch **bout)
{
	if (!slab || !L || n_regions < 0 || n_reads < 0 || (flags & ~(IHP_SLAB2_REF_2BIT | IHP_SLAB2_BASES_2BIT))) return IHP_E_ARG;
	const char *h = (const char *)slab;
	{
		// as for the first slab form: the layout is checked against ihp_slab2_layout_for before any offset is followed
		ihp_slab2_layout X;
		if (ihp_slab2_layout_for(n_regions, n_reads, 0, 0, flags, &X)) return IHP_E_ARG;
		if (L->region_read_off != X.region_read_off || L->region_base_off != X.region_base_off || L->ref_off != X.ref_off || L->ref_origin != X.ref_origin ||
		    L->start_rel != X.start_rel || L->len != X.len || L->span != X.span || L->trim_lo != X.trim_lo || L->trim_hi != X.trim_hi || L->mapq != X.mapq ||
		    L->rflags != X.rflags || L->ref_packed != X.ref_packed || L->bytes < X.bytes) return IHP_E_ARG;
		const int64_t n_bases = n_regions ? ((const int64_t *)(h + L->region_base_off))[n_regions] : 0, n_ref = n_regions ? ((const int64_t *)(h + L->ref_off))[n_regions] : 0;
		if (n_bases < 0 || n_ref < 0 || (n_reads && !n_regions) || ihp_slab2_layout_for(n_regions, n_reads, n_bases, n_ref, flags, &X)) return IHP_E_ARG;
		if (L->bases4 != X.bases4 || L->bytes != X.bytes) return IHP_E_ARG;
	}
	ihp_batch_in in;
	memset(&in, 0, sizeof(in));
	in.n_regions = n_regions; in.n_reads = n_reads;
	in.region_read_off = (const int64_t *)(h + L->region_read_off);
	in.ref_off = (const int64_t *)(h + L->ref_off); in.ref_origin = (const int64_t *)(h + L->ref_origin);
	Slab2In s2 = {slab, L, flags, (const int64_t *)(h + L->region_base_off), (const uint16_t *)(h + L->len)};
	return batch_upload_common(p, &in, slab, nullptr, bout, &s2);
}

extern "C" int ihp_batch_upload_slab(const ihp_params *p, int32_t n_regions, int64_t n_reads, const void *slab, const ihp_slab_layout *L,
                                     int32_t flags, ihp_batch **bout)
{
	if (!slab || !L || n_regions < 0 || n_reads < 0 || (flags & ~IHP_SLAB_HAS_SKIP)) return IHP_E_ARG;
	const char *h = (const char *)slab;
	{
		// the layout is the caller's word for where everything is: checked against ihp_slab_layout_for before any offset is
		// followed -- first the sections whose places depend on the counts alone (the offset arrays among them), then, with
		// the base and window totals read from those, the whole of it
		ihp_slab_layout X;
		if (ihp_slab_layout_for(n_regions, n_reads, 0, 0, &X)) return IHP_E_ARG;
		if (L->region_read_off != X.region_read_off || L->read_off != X.read_off || L->read_start != X.read_start || L->read_stop != X.read_stop ||
		    L->ref_off != X.ref_off || L->ref_origin != X.ref_origin || L->trim_lo != X.trim_lo || L->trim_hi != X.trim_hi || L->mapq != X.mapq ||
		    L->read_skip != X.read_skip || L->ref_bases != X.ref_bases || L->bytes < X.bytes) return IHP_E_ARG;
		const int64_t n_bases = n_reads ? ((const int64_t *)(h + L->read_off))[n_reads] : 0, n_ref = n_regions ? ((const int64_t *)(h + L->ref_off))[n_regions] : 0;
		if (n_bases < 0 || n_ref < 0 || ihp_slab_layout_for(n_regions, n_reads, n_bases, n_ref, &X)) return IHP_E_ARG;
		if (L->bases4 != X.bases4 || L->bytes != X.bytes) return IHP_E_ARG;
	}
	ihp_batch_in in;
	memset(&in, 0, sizeof(in));
	in.n_regions = n_regions; in.n_reads = n_reads;
	in.region_read_off = (const int64_t *)(h + L->region_read_off); in.read_off = (const int64_t *)(h + L->read_off);
	in.read_start = (const int64_t *)(h + L->read_start); in.read_stop = (const int64_t *)(h + L->read_stop);
	in.mapq = (const uint8_t *)(h + L->mapq); in.read_skip = (flags & IHP_SLAB_HAS_SKIP) ? (const uint8_t *)(h + L->read_skip) : nullptr;
	in.ref_off = (const int64_t *)(h + L->ref_off); in.ref_bases = (const uint8_t *)(h + L->ref_bases);
	in.ref_origin = (const int64_t *)(h + L->ref_origin);
	in.trim_lo = (const int32_t *)(h + L->trim_lo); in.trim_hi = (const int32_t *)(h + L->trim_hi);
	return batch_upload_common(p, &in, slab, L, bout);
}

static int batch_upload_common(const ihp_params *p, const ihp_batch_in *in, const void *slab, const ihp_slab_layout *SL, ihp_batch **bout, const Slab2In *s2)
{
	if (!p || !in || !bout || p->struct_size != (int32_t)sizeof(ihp_params)) return IHP_E_ARG;
	if (p->K < 1 || p->K > 31 || in->n_regions < 0 || in->n_reads < 0) return IHP_E_ARG;
	if (!ksw_flags_supported(p->ksw_flag)) return IHP_E_UNSUPPORTED;
	if (p->fallback && !ksw_flags_supported(p->fb_flag)) return IHP_E_UNSUPPORTED;
	if (in->n_regions && (!in->region_read_off || !in->ref_off || !in->ref_origin)) return IHP_E_ARG;
	if (!s2 && in->n_reads && (!in->read_off || (!in->bases && !slab) || !in->read_start || !in->read_stop || !in->mapq)) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	ihp_batch *b = new (std::nothrow) ihp_batch();
	if (!b) return IHP_E_NOMEM;
	const auto T0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { if (g_knob.verbose > 1) fprintf(stderr, "[ihp] upload %-12s %7.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count()); };
#define HIPB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { delete b; return hip_fail(e_, #x, __LINE__); } } while (0)
	b->P = *p; b->R = in->n_regions; b->n_reads = in->n_reads;
	const int R = b->R; const long long NR = b->n_reads;
	static const int64_t zero2[2] = {0, 0};
	const int64_t *rro = R ? in->region_read_off : zero2, *fo = R ? in->ref_off : zero2;
	if (rro[0] != 0 || rro[R] != NR) { delete b; return IHP_E_ARG; }
	// read_off at the regions' first reads: all the host needs of it (a compact slab brings exactly that; its per-read lengths
	// become read_off on the device, k_slab_expand)
	std::vector<int64_t> rb_((size_t)R + 1, 0);
	if (s2) { for (int r = 0; r <= R; ++r) rb_[(size_t)r] = R ? s2->region_base_off[r] : 0; }
	else {
		const int64_t *ro_ = NR ? in->read_off : zero2;
		for (int r = 0; r <= R; ++r) { if (rro[r] < 0 || rro[r] > NR) { delete b; return IHP_E_ARG; } rb_[(size_t)r] = ro_[rro[r]]; }
	}
	const int64_t *rb = rb_.data();
	if (rb[0] != 0) { delete b; return IHP_E_ARG; }
	b->n_bases = rb[R]; b->n_ref = fo[R];
	for (int r = 0; r < R; ++r) {
		if (rro[r + 1] < rro[r] || fo[r + 1] < fo[r] || rb[r + 1] < rb[r]) { delete b; return IHP_E_ARG; }
		const int64_t nb = rb[r + 1] - rb[r];
		if (nb > (1 << 30) || fo[r + 1] - fo[r] > (1 << 30)) { delete b; return IHP_E_CAPACITY; }
		b->max_region_bases = std::max(b->max_region_bases, (int)nb);
		b->max_region_reads = (int)std::max<int64_t>(b->max_region_reads, std::min<int64_t>(rro[r + 1] - rro[r], 1 << 30));
		b->max_ref_len = std::max(b->max_ref_len, (int)(fo[r + 1] - fo[r]));
	}
	if (s2) {
		unsigned mx = 0;
		for (long long i = 0; i < NR; ++i) mx = std::max<unsigned>(mx, s2->len[i]);
		b->max_read_len = (int)mx;
	} else {
		const int64_t *ro = NR ? in->read_off : zero2;
		for (long long i = 0; i < NR; ++i) {
			if (ro[i + 1] < ro[i]) { delete b; return IHP_E_ARG; }
			b->max_read_len = std::max<int>(b->max_read_len, (int)std::min<int64_t>(ro[i + 1] - ro[i], 1 << 30));
		}
	}
	lap("shape");
	b->h_region_read_off.assign(rro, rro + R + 1);
	b->h_ref_origin.assign(in->ref_origin, in->ref_origin + R);
	b->stream = g_streams.get();
	b->stream2 = g_streams.get();
	if (!b->stream || !b->stream2) { delete b; snprintf(g.err, sizeof(g.err), "hipStreamCreate failed"); return IHP_E_HIP; }
	hipStream_t s = b->stream;
	std::vector<int> order;                                  // regions by assembly class (cls_list), class sizes (cls_n), hand-over record offsets
	std::vector<long long> hoff;                             // (v2_hoff): staged in one page-locked block, one copy (see `aux` below)
	int cn_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define UP(buf, ptr, bytes) do { if ((rc = b->buf.upload(ptr, (size_t)(bytes), s))) { delete b; return rc; } } while (0)
	b->has_trim = (in->trim_lo != nullptr && in->trim_hi != nullptr) || s2;     // trim() done by the stager: qualities not needed
	b->has_quals = in->quals != nullptr && !b->has_trim; b->has_skip = in->read_skip != nullptr || s2;
	if (s2) {
		// the compact slab: one copy; the per-read arrays of ihp_batch_in are made on the device (k_slab_expand, enqueued below)
		const ihp_slab2_layout *L2 = s2->L;
		if ((rc = b->in_slab.alloc((size_t)L2->bytes + 64))) { delete b; return rc; }
		HIPB(hipMemcpyAsync(b->in_slab.p, s2->slab, (size_t)L2->bytes, hipMemcpyHostToDevice, s));
		void *d = b->in_slab.p;
		b->region_read_off.view(d, L2->region_read_off, sizeof(int64_t) * (R + 1));
		b->ref_off.view(d, L2->ref_off, sizeof(int64_t) * (R + 1)); b->ref_origin.view(d, L2->ref_origin, sizeof(int64_t) * R);
		if (s2->flags & IHP_SLAB2_BASES_2BIT) { b->v2_pk.view(d, L2->bases4, sizeof(uint32_t) * (size_t)((b->n_bases >> 4) + NR + 4)); b->has_pk2 = true; }
		else { b->bases4.view(d, L2->bases4, (size_t)(b->n_bases >> 1) + NR); b->has_b4 = true; }
		if ((rc = b->bases.alloc(b->n_bases)) || (rc = b->read_off.alloc(sizeof(int64_t) * (NR + 1))) || (rc = b->read_start.alloc(sizeof(int64_t) * NR)) ||
		    (rc = b->read_stop.alloc(sizeof(int64_t) * NR)) || (rc = b->trim_lo.alloc(sizeof(int32_t) * NR)) || (rc = b->trim_hi.alloc(sizeof(int32_t) * NR)) ||
		    (rc = b->mapq.alloc(NR)) || (rc = b->read_skip.alloc(NR)) || (rc = b->ref_bases.alloc(b->n_ref))) { delete b; return rc; }
	} else if (slab) {
		// one copy for everything (the caller filled a page-locked slab: ihp_host_alloc); the read bases come 4 bits each as
		// BAM stores them, k_prepack writes the ASCII ones the byte-based kernels read
		if ((rc = b->in_slab.alloc((size_t)SL->bytes + 64))) { delete b; return rc; }
		HIPB(hipMemcpyAsync(b->in_slab.p, slab, (size_t)SL->bytes, hipMemcpyHostToDevice, s));
		void *d = b->in_slab.p;
		b->region_read_off.view(d, SL->region_read_off, sizeof(int64_t) * (R + 1)); b->read_off.view(d, SL->read_off, sizeof(int64_t) * (NR + 1));
		b->read_start.view(d, SL->read_start, sizeof(int64_t) * NR); b->read_stop.view(d, SL->read_stop, sizeof(int64_t) * NR);
		b->mapq.view(d, SL->mapq, NR); if (b->has_skip) b->read_skip.view(d, SL->read_skip, NR);
		b->ref_off.view(d, SL->ref_off, sizeof(int64_t) * (R + 1)); b->ref_bases.view(d, SL->ref_bases, b->n_ref);
		b->ref_origin.view(d, SL->ref_origin, sizeof(int64_t) * R);
		b->trim_lo.view(d, SL->trim_lo, sizeof(int32_t) * NR); b->trim_hi.view(d, SL->trim_hi, sizeof(int32_t) * NR);
		b->bases4.view(d, SL->bases4, (size_t)(b->n_bases >> 1) + NR);
		b->has_b4 = true;
		if ((rc = b->bases.alloc(b->n_bases))) { delete b; return rc; }
	} else {
	UP(region_read_off, rro, sizeof(int64_t) * (R + 1));
	UP(read_off, NR ? in->read_off : zero2, sizeof(int64_t) * (NR + 1));
	UP(bases, in->bases, b->n_bases);
	if (b->has_quals) UP(quals, in->quals, b->n_bases);
	if (b->has_trim) { UP(trim_lo, in->trim_lo, sizeof(int32_t) * NR); UP(trim_hi, in->trim_hi, sizeof(int32_t) * NR); }
	UP(read_start, in->read_start, sizeof(int64_t) * NR);
	UP(read_stop, in->read_stop, sizeof(int64_t) * NR);
	UP(mapq, in->mapq, NR);
	if (b->has_skip) UP(read_skip, in->read_skip, NR);
	UP(ref_off, fo, sizeof(int64_t) * (R + 1));
	UP(ref_bases, in->ref_bases, b->n_ref);
	UP(ref_origin, in->ref_origin, sizeof(int64_t) * R);
	}
#undef UP
	lap("copies");
	// scratch sizing
	b->stage_cap = (b->max_read_len + 15) / 16 * 16 + 16;
	b->grid_retry = grid_for(R, 2);
	{
		// First-pass LDS arena: a small arena runs more waves per CU (state 3.7 KB + arena, at most 16 waves: 128
		// VGPRs) but sends the regions that do not fit on to the second pass, which runs at the occupancy of its
		// own, larger arena.  A region needs about half its read bases (live contigs stay below ~30%; headroom
		// and relocated copies take the rest) plus the staging areas, and costs about its read bases: pick the
		// arena that minimises  sum(cost / occupancy)  over the two passes.
		std::vector<std::pair<long long, long long>> need((size_t)R);          // (bytes needed, cost)
		for (long long r = 0; r < R; ++r) {
			const long long nb = rb[r + 1] - rb[r];
			need[(size_t)r] = {nb / 2 + 2 * b->stage_cap, nb + 1};
		}
		std::sort(need.begin(), need.end());
		std::vector<double> pre((size_t)R + 1, 0.0);
		for (size_t i = 0; i < (size_t)R; ++i) pre[i + 1] = pre[i] + (double)need[i].second;
		double best_t = 1e300; long long a1 = 5120;
		for (long long a = 3072; a <= 24576; a += 512) {
			const size_t fit = (size_t)(std::upper_bound(need.begin(), need.end(), std::make_pair(a - 16, (long long)1 << 60)) - need.begin());
			const double occ1 = (double)std::max(1, std::min(16, g.max_lds / ((int)a + 3840)));
			const long long a2 = std::max<long long>(12288, std::min<long long>(2 * a, g.max_lds - 24576));
			const double occ2 = (double)std::max(1, std::min(16, g.max_lds / ((int)a2 + 8192)));
			const double t = pre[fit] / occ1 + (pre[(size_t)R] - pre[fit]) / occ2;
			if (t < best_t) { best_t = t; a1 = a; }
		}
		b->lds_arena1 = (int)a1;
		b->grid_asm = grid_for(R, std::max(1, std::min(16, g.max_lds / (b->lds_arena1 + 3840))));
		if (g_knob.asm_waves) b->grid_asm = grid_for(R, std::max(1, std::min(16, g_knob.asm_waves)));
	}
	lap("arena model");
	b->lds_arena2 = std::max(12288, std::min(2 * b->lds_arena1, g.max_lds - 24576));   // + 7.5 KB (RegionStateT<128>)
	{
		long long want = ((long long)b->max_region_bases * 4 / 5 + 4 * ((b->max_read_len + 15) / 16 * 16 + 16) + 1024 + 15) / 16 * 16;
		const long long cap = (long long)g.max_lds - 16384;          // RegionStateT<256> is ~13.7 KB of static LDS
		b->lds_arena3 = (int)std::max<long long>(std::max(16384, b->lds_arena2), std::min(want, cap));
	}
	{
		// Assembly class of every region, predicted from its read bases exactly as the kernels' own up-front test
		// does (live contig bytes stay below ~30% of the read bases): class k runs in pass k's kernel from the start,
		// on a second stream beside pass 1, instead of waiting for pass 1 to forward it.  Within a class the regions
		// with the most reads go first (their serial latency is what the launch waits for at the end).
		// The packed assembly (k_asm_reads + k_asm_combine3) also takes the read-rich regions of classes 2-4 when its
		// preconditions can hold (exact matching, at most 256 reads): they join class 1 behind the usual ones and get a
		// k_asm_reads launch of their own with a larger packed area; the byte-based passes keep what is left.
		std::vector<std::pair<long long, int>> cls[4], rich;
		const long long lim[3] = {b->lds_arena1, b->lds_arena2, b->lds_arena3};
		const bool packed_ok = !g_knob.asm_v1 && !g_knob.no_rich && p->max_mismatch == 0 && p->min_overlap_pct > 0 && p->min_overlap_pct <= 1.0;
		for (int r = 0; r < R; ++r) {
			const long long nb = rb[r + 1] - rb[r];
			const long long want = nb * 3 / 10 + 2 * b->stage_cap;
			int k = 0;
			while (k < 3 && want > lim[k]) ++k;
			const long long nr_ = rro[r + 1] - rro[r];
			if (k > 0 && packed_ok && nr_ <= V3_MAXREADS_WIDE && nb <= 120000) rich.push_back({-nb, r});
			else cls[k].push_back({-nb, r});
			if (packed_ok && nr_ > 256 && nr_ <= V3_MAXREADS_WIDE && (k == 0 || nb <= 120000)) b->n_deep++;
		}
		order.reserve((size_t)R);
		std::sort(rich.begin(), rich.end());
		for (int k = 0; k < 4; ++k) {
			std::sort(cls[k].begin(), cls[k].end());
			b->n_cls[k] = (int)cls[k].size();
			for (auto &e : cls[k]) order.push_back(e.second);
			if (k == 0) { for (auto &e : rich) order.push_back(e.second); b->n_small = b->n_cls[0]; b->n_rich = (int)rich.size(); b->n_cls[0] += b->n_rich; }
		}
		const int cn6[6] = {b->n_cls[0], b->n_cls[1], b->n_cls[2], b->n_cls[3], b->n_small, b->n_rich};
		memcpy(cn_host, cn6, sizeof(cn6));
	}
	lap("classes");
	b->grid_asm = std::min(b->grid_asm, std::max(1, b->n_cls[0]));
	b->grid_asm2 = grid_for(R, std::max(1, g.max_lds / (b->lds_arena2 + 8192)));
	b->grid_asm3 = grid_for(R, std::max(1, g.max_lds / (b->lds_arena3 + 14336)));
	{
		// Packed read phase for class 1 (asm2_dev.h, k_assemble2) when the parameters allow it: exact matching only
		// (max_mismatch 0) and 0 < min_overlap_pct <= 1.  Its byte arena holds the contigs that reach combine (no read
		// staging); the packed area holds the longest read, one record per read and the 2-bit contig slots.
		b->v2 = !g_knob.asm_v1 && p->max_mismatch == 0 && p->min_overlap_pct > 0 && p->min_overlap_pct <= 1.0 && b->n_cls[0] > 0;
		if (b->v2) {
			long long nb1 = 0, nr1 = 0, nbL = 0, nrL = 0;
			for (int r = 0; r < R; ++r) {
				const long long nb = rb[r + 1] - rb[r];
				if (nb * 3 / 10 + 2 * b->stage_cap <= b->lds_arena1) { nb1 = std::max(nb1, nb); nr1 = std::max<long long>(nr1, rro[r + 1] - rro[r]); }
				else if (b->n_rich && rro[r + 1] - rro[r] <= V3_MAXREADS_WIDE && nb <= 120000) { nbL = std::max(nbL, nb); nrL = std::max<long long>(nrL, rro[r + 1] - rro[r]); }
			}
			if (nb1 == 0) { nb1 = std::min<long long>(nbL, 16384); nr1 = std::min<long long>(nrL, 64); }   // no region of the usual size: size the first tier for small ones
			// what a region needs at least ...
			long long need_pdw = 1 + (b->max_read_len + 15) / 16 + 2 + nr1 + nb1 / 16 * (b->max_read_len > 200 ? 12 : 6) / 10 + 64 + (b->max_read_len > 200 ? 4 : 1) * V2_HB_DW + (b->max_read_len > 200 ? V2_WLX : 0);   // long reads: more single-read contigs, longer relocations
			// ... and what the occupancy that need allows leaves unused: a region that runs out of room is assembled again from
			// scratch by the byte-based passes, one serial chain of ~0.6 ms, so room is worth more than the last wave.
			const int occ_r = (int)std::max<long long>(1, std::min<long long>(32, g.max_lds / (4 * need_pdw + 256)));
			need_pdw = std::max<long long>(need_pdw, (g.max_lds / occ_r - 256) / 4);
			b->v2_pdw = g_knob.v2_pdw ? g_knob.v2_pdw : (int)need_pdw;
			b->v2_pdw = b->v2_pdw / 4 * 4;
			b->v2_nb1 = nb1;
			size_combine_tiers(b, 0);
			b->tier_occ_default = b->tier_occ;
			b->tier_sig = (b->max_read_len / 32) | ((int)std::min<long long>(nb1 / 2048, 0xffff) << 8) | (b->tier_occ_default << 24);
			const int dyn_max = g.max_lds - 8192 - 1024;
			const int per_wave = (int)comb_wave_bytes(b->v2_arena, b->tier_wide ? comb_stat() : comb_stat_a()), per_wave_r = 4 * b->v2_pdw + 256;
			if (b->v2_arena + 4 * b->v2_pm > dyn_max || per_wave > g.max_lds - 1024 || per_wave_r > g.max_lds - 1024) b->v2 = false;
			else {
				b->grid_v2r = std::min(grid_for(R, std::max(1, std::min(g_knob.asmr_waves ? g_knob.asmr_waves : 32, g.max_lds / per_wave_r))), std::max(1, b->n_cls[0]));
				b->grid_pack = grid_for((int)std::min<long long>((NR + 3) / 4, 1 << 30), 32);
				if (b->n_rich) {
					// the read-rich launch: the same kernel with the packed area its largest region needs (reads + slots of ~0.4
					// of the read bases + relocations), at the occupancy that leaves
					long long pdwL = 1 + (b->max_read_len + 15) / 16 + 2 + nrL + nbL / 16 * (b->max_read_len > 200 ? 12 : 7) / 10 + 192 + (b->max_read_len > 200 ? 4 : 1) * V2_HB_DW + (b->max_read_len > 200 ? V2_WLX : 0);
					pdwL = std::min<long long>(pdwL, (g.max_lds - 4096) / 4);
					const int occL = (int)std::max<long long>(1, std::min<long long>(32, g.max_lds / (4 * pdwL + 256)));
					pdwL = std::max<long long>(pdwL, (g.max_lds / occL - 256) / 4);
					b->v2_pdw_rich = (int)(pdwL / 4 * 4);
					b->grid_v2r_rich = std::min(grid_for(R, occL), b->n_rich);
				}
				// hand-over records between the two kernels: 8 + 9 min(64, reads) + reads + bases / 16 + 8 dwords per region
				hoff.assign((size_t)R + 1, 0);
				for (int r = 0; r < R; ++r) {
					const long long nr = rro[r + 1] - rro[r], nb = rb[r + 1] - rb[r];
					hoff[(size_t)r + 1] = hoff[(size_t)r] + ((16 + 9 * std::min<long long>(64, nr) + nr + nb / 16 + 3) / 4 * 4);
				}
				b->v2_hand_dwords = hoff[(size_t)R];
			}
		}
	}
	{
		// Run-time overflow launches (a region that ran out of arena / contig slots in the pass its read bases
		// predicted).  When every region of the batch was predicted to fit pass 1 these lists are almost always
		// empty, and a full persistent grid of large-LDS workgroups costs 20-30 us per empty launch: two workgroups
		// per CU then.  Batches with read-rich classes get the full grids (their lists are well used).
		const bool side = b->n_cls[1] + b->n_cls[2] + b->n_cls[3] > 0;
		const int small = 2 * g.cus;
		b->grid_ovf1 = std::min(b->grid_asm, small);           // regions the packed pass hands back to the byte-based class-1 kernel
		b->grid_ovf2 = side ? b->grid_asm2 : std::min(b->grid_asm2, small);
		b->grid_ovf3 = side ? b->grid_asm3 : std::min(b->grid_asm3, small);
		b->grid_ovf4 = side ? b->grid_retry : std::min(b->grid_retry, small);
	}
	b->arena_cap = (3 * b->max_region_bases + 4 * b->stage_cap + 2048 + 15) / 16 * 16;
	b->corr_cap = std::min(MAXLEN, b->max_region_bases) + 16;
	const long long slots = NR;
	// ksw2: contig length is bounded by the region's read bases (every contig base comes from a read)
	const int qmax = std::min(MAXLEN, b->max_region_bases), tmax = b->max_ref_len;
	{
		const bool fastp = p->bw >= 0 && p->bw <= 62;
		size_t need = (fastp ? ksw_narrow_lds_bytes(qmax, tmax) : std::max(ksw_lds_bytes(qmax, tmax), ksw_wide_lds_bytes<6>(qmax, tmax))) + 64;
		// contigs are rarely longer than the reference window + band; cap the LDS request there and let the
		// kernel flag anything larger (reported as IHP_E_CAPACITY for that batch)
		const size_t typical = (fastp ? ksw_narrow_lds_bytes(tmax + 64, tmax) : std::max(ksw_lds_bytes(tmax + 64, tmax), ksw_wide_lds_bytes<6>(tmax + 64, tmax))) + 64;
		need = std::min(need, std::max(typical, (size_t)(fastp ? 2048 : 8192)));
		need = std::min(need, (size_t)g.max_lds - 2048);
		b->lds_ksw = (int)need;
		const int w = p->bw < 0 ? std::max(qmax, tmax) : p->bw;
		const int nc = (std::min(std::min(qmax, tmax), w + 1) + 15) / 16 + 1;
		const int qeff = std::min(qmax, tmax + 2 * std::max(w, 64));
		b->p_cap = std::max(((size_t)(qeff + tmax) * nc + 1) * 16, ksw_narrow_p_bytes(qeff, tmax)) + 64;
		b->cig_cap = qeff + tmax + 8;
		// the pair launch (ksw_pair.h) takes contigs at least w + 1 bases shorter than their window
		b->lds_ksw_pair = 0; b->p_cap_pair = 0;
		const int qpair = std::min(qmax, tmax - p->bw - 1);
		if (fastp && p->bw >= 49 && qpair >= p->bw + 32) {
			b->lds_ksw_pair = (int)std::min(ksw_pair_lds_bytes(qpair, tmax, tmax) + 64, (size_t)g.max_lds - 2048);
			b->p_cap_pair = ksw_pair_p_bytes(qpair, p->bw) + 64;
		}
		// the roomy launch for the jobs that do not fit that (a contig much longer than its window): a few workgroups, all the LDS,
		// scratch for the longest contig the assembly can leave (up to 256 MB each; what still does not fit is IHP_E_CAPACITY)
		const int ncq = (std::min(std::min(qmax, tmax), w + 1) + 15) / 16 + 1;
		b->p_cap_big = std::min<size_t>(std::max(((size_t)(qmax + tmax) * ncq + 1) * 16, ksw_narrow_p_bytes(qmax, tmax)) + 64, (size_t)256 << 20);
		b->cig_cap_big = qmax + tmax + 8;
		b->grid_kovf = (int)std::max<size_t>(1, std::min<size_t>(16, ((size_t)512 << 20) / b->p_cap_big));
	}
	b->grid_ksw = grid_for((int)std::min<long long>(slots, 1 << 30), g_knob.ksw_waves ? g_knob.ksw_waves : 32);
	b->grid_tally = grid_for((int)std::min<long long>(slots, 1 << 30), g_knob.tally_waves ? g_knob.tally_waves : 32);
	const long long njobs_cap = std::min<long long>(slots, (long long)R * std::max(1, p->max_pre_contigs));
	b->njobs_cap = njobs_cap;
	b->cig_bump_cap = 8 * njobs_cap + 4096;                      // CIGARs longer than CIG_SLOT words
	b->cig_pool_cap = b->cig_bump_cap + (long long)CIG_SLOT * njobs_cap;
	b->ev_pool_cap = (long long)std::max(1, p->max_events) * njobs_cap + 16;
	if (g_limits[0] > 0) { b->cig_bump_cap = std::min(b->cig_bump_cap, g_limits[0]); b->cig_pool_cap = b->cig_bump_cap + (long long)CIG_SLOT * njobs_cap; }
	if (g_limits[1] > 0) b->ev_pool_cap = std::min(b->ev_pool_cap, g_limits[1]);
	if (g_limits[3] > 0) { b->p_cap = std::min(b->p_cap, (size_t)g_limits[3]); b->p_cap_big = std::min(b->p_cap_big, (size_t)g_limits[3]); }
	if (g_knob.ksw_p_cap > 0) b->p_cap = std::min(b->p_cap, (size_t)g_knob.ksw_p_cap);
	if (g_limits[3] > 0 || g_knob.ksw_p_cap > 0) b->p_cap_pair = std::min(b->p_cap_pair, 2 * b->p_cap);
	if (p->fallback) {
		// a read against the rest of the reference window / of the contig from the read's start.  Contigs are rarely
		// longer than the window; the scratch is sized for that and the kernel flags anything larger (IHP_E_CAPACITY).
		const int ql = std::max(1, b->max_read_len), tl = std::max(1, std::min(qmax, tmax + 128));
		const int tt = std::max(tl, tmax);
		const int w = p->fb_bw < 0 ? std::max(ql, tt) : p->fb_bw;
		const int nc = (std::min(std::min(ql, tt), w + 1) + 15) / 16 + 1;
		b->fb_p_cap = ((size_t)(ql + tt) * nc + 1) * 16 + 64;
		b->fb_cig_cap = ql + tt + 16;
		int8_t fmat[25];
		ihp_matrix(p->fb_match, p->fb_mismatch, fmat);
		const KswParams FP = make_ksw_params(5, fmat, (int8_t)std::abs((int)p->fb_gap_open), (int8_t)std::abs((int)p->fb_gap_ext),
		                                     p->fb_bw, p->fb_zdrop, p->fb_flag, 1);
		size_t lneed = ksw_lds_bytes(ql, tt);                        // the generic LDS sweep always fits this
		if (!(p->fb_flag & KSW_EZ_RIGHT) && ksw_wide_ok<3>(FP, ql, tt)) lneed = std::max(ksw_wide_lds_bytes<3>(ql, tt), ksw_lds_bytes(std::min(ql, 32), tt));
		else if (!(p->fb_flag & KSW_EZ_RIGHT) && ksw_wide_ok<6>(FP, ql, tt)) lneed = std::max(ksw_wide_lds_bytes<6>(ql, tt), ksw_lds_bytes(std::min(ql, 32), tt));
		if (ksw_duo_ok(FP, std::min(ql, 64 * DUO_NS_MAX), tt, tt)) {    // both alignments of an item in one sweep (ksw_duo.h)
			lneed = std::max(lneed, ksw_duo_lds_bytes(tt));
			b->fb_p_cap = std::max(b->fb_p_cap, ksw_duo_p_bytes(std::min(ql, 64 * DUO_NS_MAX), tt));
		}
		b->lds_fb = (int)std::min<size_t>(lneed + 64, (size_t)g.max_lds - 2048);
		b->grid_fb = grid_for(1 << 30, std::max(1, std::min(16, g.max_lds / (b->lds_fb + 256))));
		// the roomy launch for the items that do not fit that (a read of an event on a contig far longer than its window): the roomy
		// ksw2 launch's few workgroups and its scratch, cut for the longest contig the assembly can leave (up to 256 MB each)
		{
			const int tbig = std::max(qmax, tmax);
			const int wb = p->fb_bw < 0 ? std::max(ql, tbig) : p->fb_bw;
			const int ncb = (std::min(std::min(ql, tbig), wb + 1) + 15) / 16 + 1;
			size_t pb = ((size_t)(ql + tbig) * ncb + 1) * 16 + 64;
			if (ksw_duo_ok(FP, std::min(ql, 64 * DUO_NS_MAX), tbig, tbig)) pb = std::max(pb, ksw_duo_p_bytes(std::min(ql, 64 * DUO_NS_MAX), tbig));
			b->fb_p_cap_big = std::min<size_t>(pb, (size_t)256 << 20);
			b->fb_cig_cap_big = ql + tbig + 16;
			const size_t pall = std::max(b->p_cap_big, b->fb_p_cap_big);
			b->grid_kovf = (int)std::max<size_t>(1, std::min<size_t>((size_t)(b->grid_kovf > 0 ? b->grid_kovf : 16), ((size_t)512 << 20) / pall));
		}
		if (g_knob.fb_p_cap > 0) b->fb_p_cap = std::min(b->fb_p_cap, (size_t)g_knob.fb_p_cap);
	}
	lap("sizing");
	static_assert(sizeof(int) * M_WORDS <= ihp_batch::Z_TIMES, "misc counters overlap the stamps");
	if ((rc = b->misc.alloc(b->z_bytes())) || (rc = b->summary.alloc(sizeof(ihp_region_summary) * (size_t)R))) { delete b; return rc; }
	b->hit_cap = 2 * (2 * HIT_SLOTS * NR) + 128 * (long long)std::max(1, b->max_region_reads);
	if (g_limits[2] > 0) b->hit_cap = std::min(b->hit_cap, std::max(g_limits[2], 2 * HIT_SLOTS * NR));   // the fixed slots stay; the bump region shrinks
	if (b->v2 || b->has_b4 || b->has_pk2) {
		b->grid_pack = grid_for((int)std::min<long long>((NR + 3) / 4, 1 << 30), 32);
		if ((!b->has_pk2 && (rc = b->v2_pk.alloc(sizeof(uint32_t) * (size_t)((b->n_bases >> 4) + NR + 4)))) || (rc = b->v2_trim_lo.alloc(sizeof(int) * (size_t)NR)) ||
		    (rc = b->v2_trim_hi.alloc(sizeof(int) * (size_t)NR)) || (rc = b->v2_read_bad.alloc((size_t)NR))) { delete b; return rc; }
	}
	{
		// the small arrays built above travel in one copy from a page-locked block the batch keeps until it is freed:
		// no wait in here for a copy out of a local
		const size_t o_cn = ((size_t)R * sizeof(int) + 63) / 64 * 64, o_hoff = o_cn + 64;
		const size_t total = o_hoff + (hoff.empty() ? 0 : sizeof(long long) * hoff.size());
		b->aux_host = g_slabs.get(total);
		if (!b->aux_host) { delete b; snprintf(g.err, sizeof(g.err), "hipHostMalloc of %zu bytes failed", total); return IHP_E_NOMEM; }
		char *h = (char *)b->aux_host + sizeof(SlabHdr);
		if (R) memcpy(h, order.data(), sizeof(int) * (size_t)R);
		memcpy(h + o_cn, cn_host, sizeof(cn_host));
		if (!hoff.empty()) memcpy(h + o_hoff, hoff.data(), sizeof(long long) * hoff.size());
		if ((rc = b->aux.alloc(total))) { delete b; return rc; }
		HIPB(hipMemcpyAsync(b->aux.p, h, total, hipMemcpyHostToDevice, s));
		b->cls_list.view(b->aux.p, 0, sizeof(int) * (size_t)R); b->cls_n.view(b->aux.p, o_cn, sizeof(cn_host));
		if (!hoff.empty()) b->v2_hoff.view(b->aux.p, o_hoff, sizeof(long long) * hoff.size());
	}
	lap("aux");
	if ((rc = alloc_work(b))) { delete b; return rc; }
	lap("alloc_work");
	for (auto &e : b->ev) HIPB(hipEventCreate(&e));
	HIPB(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_bfork, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_bjoin, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_kfork, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_kjoin, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_rfork, hipEventDisableTiming));
	HIPB(hipEventCreateWithFlags(&b->ev_rjoin, hipEventDisableTiming));
	lap("events");
	b->report = g_reports.get();
	if (!b->report) { delete b; snprintf(g.err, sizeof(g.err), "hipHostMalloc of the report page failed"); return IHP_E_NOMEM; }
	HIPB(hipMemsetAsync(b->misc.p, 0, b->z_bytes(), s));      // the only memset of the batch's life (see ihp_batch::misc)
	if (s2) {
		const ihp_slab2_layout *L2 = s2->L;
		const char *d = (const char *)b->in_slab.p;
		SlabExpandArgs x;
		x.n_regions = R; x.n_reads = NR;
		x.region_read_off = (const long long *)(d + L2->region_read_off); x.region_base_off = (const long long *)(d + L2->region_base_off);
		x.ref_off = (const long long *)(d + L2->ref_off); x.ref_origin = (const long long *)(d + L2->ref_origin);
		x.start_rel = (const int *)(d + L2->start_rel); x.len = (const unsigned short *)(d + L2->len); x.span = (const unsigned short *)(d + L2->span);
		x.trim_lo_in = (const unsigned short *)(d + L2->trim_lo); x.trim_hi_in = (const unsigned short *)(d + L2->trim_hi);
		x.mapq_in = (const uint8_t *)(d + L2->mapq); x.rflags = (const uint8_t *)(d + L2->rflags);
		x.ref_packed = (const uint8_t *)(d + L2->ref_packed); x.ref_2bit = (s2->flags & IHP_SLAB2_REF_2BIT) ? 1 : 0;
		x.read_off = b->read_off.as<long long>(); x.read_start = b->read_start.as<long long>(); x.read_stop = b->read_stop.as<long long>();
		x.trim_lo = b->trim_lo.as<int>(); x.trim_hi = b->trim_hi.as<int>(); x.mapq = b->mapq.as<uint8_t>(); x.read_skip = b->read_skip.as<uint8_t>();
		x.ref_bases = b->ref_bases.as<uint8_t>(); x.bad = b->misc.as<int>() + M_SLAB_BAD;
		hipLaunchKernelGGL(k_slab_expand, dim3(std::max(1, grid_for(std::max(R, 1), 32))), dim3(64), 0, s, x);
		HIPB(hipGetLastError());
	}
	// separate arrays may change once this returns; a slab stays as it is until a wait for the batch has returned
	// (ihp_batch_sync, fetch, ...: see ihp_batch_upload_slab in the header), so its copy is left in flight
	if (!slab) HIPB(hipStreamSynchronize(s));
#undef HIPB
	lap("end");
	*bout = b;
	return 0;
}

static int pack_counts_enqueue(ihp_batch *b);

extern "C" int ihp_batch_run(ihp_batch *b)
{
	if (!b) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	hipStream_t s = b->stream;
	const ihp_params &p = b->P;
	{ int rc1 = alloc_work(b); if (rc1) return rc1; }
	// the shape of the batch: what its plan is remembered under
	{
		const long long per_region = b->R > 0 ? b->n_bases / b->R : 0;
		unsigned long long k = (unsigned long long)(b->max_read_len / 32) | (unsigned long long)std::min<long long>(per_region / 2048, 0xffff) << 12;
		k |= (unsigned long long)(b->v2 ? 1 : 0) << 28 | (unsigned long long)(b->tier_occ_default & 63) << 29 | (unsigned long long)(p.max_mismatch & 15) << 35;
		k |= (unsigned long long)(p.bw & 0xff) << 39 | (unsigned long long)(p.fallback ? 1 : 0) << 47 | (unsigned long long)(b->max_ref_len / 64 & 0xfff) << 48;
		b->hint_key = k | 1ull << 63;
	}
	TierHint H;
	const bool have_hint = g_hints.get(b->hint_key, H) && !g_knob.no_hint;
	// the roomy ksw2 launch (below) is left out when the last batch of this shape had no job for it; its scratch -- up to 512 MB,
	// a pool miss is a hipMalloc, which synchronises the device -- is taken HERE, before anything of this run is enqueued: a
	// failure returns with nothing in the stream
	const bool ksw_roomy = b->R > 0 && b->n_reads > 0 && !(have_hint && !g_knob.no_spec && !b->force_full && H.n_kovf == 0 && H.clean_k >= CLEAN_MIN);
	if (ksw_roomy) {
		// (shared with the roomy launch of the alignment fallback, which runs behind the tally)
		if (!b->p_scratch_big.p) { int rcb = b->p_scratch_big.alloc(std::max(b->p_cap_big, b->fb_p_cap_big) * std::max(1, b->grid_kovf)); if (rcb) return rcb; }
		if (!b->cig_tmp_big.p) { int rcb = b->cig_tmp_big.alloc(sizeof(uint32_t) * (size_t)std::max(b->cig_cap_big, b->fb_cig_cap_big) * std::max(1, b->grid_kovf)); if (rcb) { b->p_scratch_big.release(); return rcb; } }
	}
	// counters, stamps, work queues and per-region hit counts are zero: cleared at upload and by the previous run's k_summary
	int *wq = b->queues_dev();
	const bool profiling = g_knob.profile != 0;
	if (b->dirty) {                                                         // the previous run ended before its k_summary was enqueued
		// (everything but M_SLAB_BAD: k_slab_expand raises it ONCE, at upload, and it has to reach the first wait however many runs
		// -- cut short or not -- lie in between)
		HIPC(hipMemsetAsync(b->misc.p, 0, sizeof(int) * M_SLAB_BAD, s));
		HIPC(hipMemsetAsync((char *)b->misc.p + sizeof(int) * (M_SLAB_BAD + 1), 0, b->z_bytes() - sizeof(int) * (M_SLAB_BAD + 1), s));
	}
	b->dirty = true;
	if (profiling) HIPC(hipMemsetAsync(b->prof.p, 0, sizeof(long long) * 64, s));
	int *misc = b->misc.as<int>();
	unsigned long long *tm = b->timing ? b->times_dev() : nullptr;
	HIPC(hipEventRecord(b->ev[0], s));
	bool spec_skipped_run = false, ksw_skipped_run = false, fb_roomy_skipped_run = false;
	if (b->R > 0) {
		AsmArgs a;
		a.n_regions = b->R;
		a.region_read_off = b->region_read_off.as<long long>(); a.read_off = b->read_off.as<long long>();
		a.bases = b->bases.as<uint8_t>(); a.quals = b->has_quals ? b->quals.as<uint8_t>() : nullptr;
		a.trim_lo = b->has_trim ? b->trim_lo.as<int>() : nullptr; a.trim_hi = b->has_trim ? b->trim_hi.as<int>() : nullptr;
		a.read_start = b->read_start.as<long long>(); a.read_stop = b->read_stop.as<long long>();
		a.mapq = b->mapq.as<uint8_t>(); a.read_skip = b->has_skip ? b->read_skip.as<uint8_t>() : nullptr;
		a.ref_off = b->ref_off.as<long long>(); a.ref_origin = b->ref_origin.as<long long>();
		a.min_overlap_pct = p.min_overlap_pct; a.min_mapq_assemble = p.min_mapq_assemble; a.min_mapq_stop = p.min_mapq_stop;
		a.trim_min_qual = p.trim_min_qual; a.combine_min_support = p.combine_min_support;
		a.combine_min_overlap = p.combine_min_overlap; a.max_mismatch = p.max_mismatch;
		a.max_pre_contigs = p.max_pre_contigs; a.min_ctg_len = p.min_ctg_len; a.min_reads = p.min_reads;
		a.K = p.K; a.ref_pad = p.ref_pad;
		a.stage_cap = b->stage_cap; a.corr = b->corr.as<Corr>(); a.corr_cap = b->corr_cap;
		a.status = b->status.as<int>(); a.n_pre = b->n_pre.as<int>(); a.n_final = b->n_final.as<int>();
		a.ctg_start = b->ctg_start.as<long long>(); a.ctg_nreads = b->ctg_nreads.as<long long>();
		a.ctg_seq_off = b->ctg_seq_off.as<long long>(); a.ctg_len = b->ctg_len.as<int>();
		a.aln_flags = b->aln_flags.as<int>(); a.aln_ref_len = b->aln_ref_len.as<int>();
		a.aln_ref_start = b->aln_ref_start.as<long long>();
		a.out_seq = b->out_seq.as<uint8_t>(); a.out_sup = b->out_sup.as<uint32_t>();
		a.jobs = b->jobs.as<AlnJob>(); a.n_jobs = misc + M_NJOBS; a.work_counter = wq;
		a.prof = profiling ? b->prof.as<long long>() : nullptr;
		a.t_start = nullptr;
		a.v2_pk = nullptr; a.v2_trim_lo = a.v2_trim_hi = nullptr; a.v2_read_bad = nullptr; a.v2_pdw = 0; a.v2_pm_dw = 0; a.v2_hand = nullptr; a.v2_hoff = nullptr; a.lpt_cnt = nullptr; a.lpt_seg = nullptr; a.lpt_stride = 0; a.lpt_nclass = 0;
		// Passes 1-3: LDS arenas of growing size (falling occupancy); pass 4: HBM arena (catch-all).  Every region
		// starts in the pass its read bases predict (classes built at upload): class 1 on the batch stream, classes
		// 2-4 one after the other on a second stream beside it, so the long serial latency of the read-rich regions
		// overlaps pass 1 instead of following it.  A region that still runs out of arena / contig slots is
		// forwarded at run time to the next pass's overflow list; those lists (empty on pile-ups like C2, well used
		// when long reads with errors leave many single-read contigs) are processed after both streams have joined.
		int *o2 = b->retry_list.as<int>(), *o3 = b->retry_list2.as<int>(), *o4 = b->retry_list3.as<int>();
		bool spec_skip = false;
		const int *cl = b->cls_list.as<int>(), *cn = b->cls_n.as<int>();
		const int n1 = b->n_cls[0], n2 = b->n_cls[1], n3 = b->n_cls[2], n4 = b->n_cls[3];
		hipStream_t s2 = b->stream2;
		const bool side = n2 + n3 + n4 > 0;
		// The first combine tier is sized from the read bases of the usual region (size_combine_tiers); when the last batch of
		// this shape filed most of its regions above that, the tiers are cut again so that the first one holds 85 % of them.
		if (b->v2 && have_hint && !g_knob.comb_occ && !g_knob.v2_arena && H.sig == b->tier_sig && H.regions > 0) {
			const long long tot = H.regions;
			// more than a hundredth of the regions with more contigs than the short table holds: the first tier runs the build with
			// the full table (C3, C5: 6-7 % -- a second tier for them alone, at two thirds of the occupancy, every step)
			const bool wide = (long long)H.n_manyc * 100 > tot;
			// a tier of its own for a few percent of the regions costs a round of the heaviest ones at the end: when a first tier
			// of not much lower occupancy holds (nearly) all regions -- a narrow distribution just above the predicted arena --
			// it is taken; otherwise (regions of very different sizes) the first tier is cut for 85 % and the others take the rest.
			// The histogram's capacities are those of the build the last batch ran (H.wide); when the build changes the tiers
			// are first cut for it from the read bases and settle with the batch after.
			const int occ_lim = 21;                                  // (above 20-21 regions per CU the launch gains nothing more)
			int want = 0, want_all = 0;
			long long cum = 0;
			for (int k = 0; k < HIST_N; ++k) {
				cum += H.hist[k];
				if (HIST_OCC[k] > occ_lim) continue;
				if (!want && cum * 100 >= 85 * tot) want = HIST_OCC[k];
				if (!want_all && cum * 200 >= 199 * tot) want_all = HIST_OCC[k];
			}
			if (!want) want = HIST_OCC[HIST_N - 1];
			if (want_all && want_all * 10 >= want * 7) want = want_all;
			if (wide != b->tier_wide) { b->tier_wide = wide; size_combine_tiers(b, wide == (H.wide != 0) ? want : 0); }
			else if (wide == (H.wide != 0) && want != b->tier_occ) size_combine_tiers(b, want);
		}
		if (g_knob.verbose && b->v2 && have_hint)
			fprintf(stderr, "[ihp] hist (regions whose contigs fit a first tier of 25 21 18 16 14 12 11 10 9 8 7 waves/CU and no higher one): %d %d %d %d %d %d %d %d %d %d %d of %d\n",
			        H.hist[0], H.hist[1], H.hist[2], H.hist[3], H.hist[4], H.hist[5], H.hist[6], H.hist[7], H.hist[8], H.hist[9], H.hist[10], H.regions);
		if (g_knob.verbose && b->v2 && have_hint) fprintf(stderr, "[ihp] %d regions with more than %d contigs; first tier's contig table: %s\n", H.n_manyc, COMB_MAXC_A, b->tier_wide ? "full" : "short");
		if (g_knob.verbose && b->v2)
			fprintf(stderr, "[ihp] run: %d regions, first tier %d waves/CU (default %d): arenas %d / %d / %d / %d, grids %d / %d / %d / %d; hint valid %d b %d c %d big %d back %d\n",
			        b->R, b->tier_occ, b->tier_occ_default, b->v2_arena, b->v2_arena_b, b->v2_arena_c, b->v2_arena_big, b->grid_v2, b->grid_v2b, b->grid_v2c, b->grid_v2big,
			        (int)have_hint, H.n_b, H.n_c, H.n_big, H.n_back);
		auto pass2 = [&](AsmArgs x, hipStream_t st, const int *in, const int *n_in, int set, Corr *corr, int grid) {
			x.arena_seq = nullptr; x.arena_sup = b->lds_sup2.as<uint32_t>(); x.arena_cap = b->lds_arena2; x.lds_arena = b->lds_arena2;
			x.in_list = in; x.n_in = n_in; x.out_list = o3; x.n_out = misc + M_NRETRY2; x.work_counter = wq + set * WQ_WORDS; x.corr = corr;
			hipLaunchKernelGGL((k_assemble<128, true, 1>), dim3(grid), dim3(64), b->lds_arena2, st, x);
		};
		auto pass3 = [&](AsmArgs x, hipStream_t st, const int *in, const int *n_in, int set, Corr *corr, int grid) {
			x.arena_seq = nullptr; x.arena_sup = b->lds_sup3.as<uint32_t>(); x.arena_cap = b->lds_arena3; x.lds_arena = b->lds_arena3;
			x.in_list = in; x.n_in = n_in; x.out_list = o4; x.n_out = misc + M_NRETRY3; x.work_counter = wq + set * WQ_WORDS; x.corr = corr;
			hipLaunchKernelGGL((k_assemble<256, true, 1>), dim3(grid), dim3(64), b->lds_arena3, st, x);
		};
		auto pass4 = [&](AsmArgs x, hipStream_t st, const int *in, const int *n_in, int set, Corr *corr, int grid) {
			x.arena_seq = b->arena_seq.as<uint8_t>(); x.arena_sup = b->arena_sup.as<uint32_t>(); x.arena_cap = b->arena_cap; x.lds_arena = 0;
			x.in_list = in; x.n_in = n_in; x.out_list = nullptr; x.n_out = nullptr; x.work_counter = wq + set * WQ_WORDS; x.corr = corr;
			hipLaunchKernelGGL((k_assemble<1024, false, 1>), dim3(grid), dim3(64), 0, st, x);
		};
		// k_prepack first: the 2-bit reads, the kept ranges and the "not ACGT" flags of every read (and, for a batch that came
		// with 4-bit bases, the ASCII bases every other kernel reads)
		const bool prepacked = (n1 && b->v2) || b->has_b4 || b->has_pk2;
		PrepackArgs pa;
		memset(&pa, 0, sizeof(pa));
		if (prepacked) {
			pa.n_reads = b->n_reads; pa.read_off = b->read_off.as<long long>(); pa.bases = b->bases.as<uint8_t>();
			pa.bases4 = b->has_b4 ? b->bases4.as<uint8_t>() : nullptr; pa.bases_w = b->bases.as<uint8_t>();
			pa.quals = b->has_quals ? b->quals.as<uint8_t>() : nullptr;
			pa.trim_lo_in = b->has_trim ? b->trim_lo.as<int>() : nullptr; pa.trim_hi_in = b->has_trim ? b->trim_hi.as<int>() : nullptr;
			pa.trim_min_qual = p.trim_min_qual; pa.pk = b->v2_pk.as<uint32_t>(); pa.trim_lo = b->v2_trim_lo.as<int>();
			pa.trim_hi = b->v2_trim_hi.as<int>(); pa.read_bad = b->v2_read_bad.as<uint8_t>();
			pa.t_start = tm;                                       // the first launch of the stage
			// (the pipelined kernel loads without asking whether a read has bases: not for a batch without any)
			const int pf = (pa.trim_lo_in && pa.n_reads > 0 && b->n_bases >= 16) ? g_knob.prepack_fast : 0;
			if (b->has_pk2) hipLaunchKernelGGL(k_unpack_pk, dim3(b->grid_pack), dim3(64), 0, s, pa);     // (the slab brought the packed form: only the ASCII copy, the kept ranges and the flags are left to write)
			else if (pf > 0 && pa.bases4) {                             // (a slab: 4-bit bases in, their ASCII form written on the way)
				if (pf == 1) hipLaunchKernelGGL((k_prepack_fast<1, true>), dim3(b->grid_pack), dim3(64), 0, s, pa);
				else hipLaunchKernelGGL((k_prepack_fast<2, true>), dim3(b->grid_pack), dim3(64), 0, s, pa);
			}
			else if (pf == 1) hipLaunchKernelGGL(k_prepack_fast<1>, dim3(b->grid_pack), dim3(64), 0, s, pa);
			// (88 registers: 20 waves of this build are resident on a CU, and its waves all do the same work -- a grid of 32 per CU ran as
			// one full round and one of twelve waves)
			else if (pf == 2) hipLaunchKernelGGL(k_prepack_fast<2>, dim3(std::min(b->grid_pack, g.cus * 20)), dim3(64), 0, s, pa);
			else if (pf == 3) hipLaunchKernelGGL(k_prepack_fast<3>, dim3(b->grid_pack), dim3(64), 0, s, pa);
			else if (pf == 4) hipLaunchKernelGGL(k_prepack_fast<4>, dim3(b->grid_pack), dim3(64), 0, s, pa);
			else hipLaunchKernelGGL(k_prepack, dim3(b->grid_pack), dim3(64), 0, s, pa);
			HIPC(hipGetLastError());
		}
		if (side) {
			HIPC(hipEventRecord(b->ev_fork, s));                   // the counters are cleared
			HIPC(hipStreamWaitEvent(s2, b->ev_fork, 0));
			if (n2) pass2(a, s2, cl + n1, cn + 1, 1, b->corr2.as<Corr>(), b->grid_asm2);
			if (n3) pass3(a, s2, cl + n1 + n2, cn + 2, 2, b->corr2.as<Corr>(), b->grid_asm3);
			if (n4) pass4(a, s2, cl + n1 + n2 + n3, cn + 3, 3, b->corr2.as<Corr>(), b->grid_retry);
			HIPC(hipGetLastError());
			HIPC(hipEventRecord(b->ev_join, s2));
		}
		if (n1 && b->v2) {
			// class 1 through the packed read phase; what it cannot take goes to the byte-based class-1 kernel behind it
			AsmArgs x = a;
			x.v2_pk = pa.pk; x.v2_trim_lo = pa.trim_lo; x.v2_trim_hi = pa.trim_hi; x.v2_read_bad = pa.read_bad; x.v2_pdw = b->v2_pdw;
			x.v2_hand = b->v2_hand.as<uint32_t>(); x.v2_hoff = b->v2_hoff.as<long long>();
			x.arena_seq = nullptr; x.arena_sup = nullptr; x.arena_cap = b->v2_arena; x.lds_arena = b->v2_arena;
			x.in_list = cl; x.n_in = cn; x.out_list = b->retry_list0.as<int>(); x.n_out = misc + M_NRETRY0; x.work_counter = wq + 10 * WQ_WORDS;
			x.t_start = nullptr;
			ReadArgs ra;
			ra.region_read_off = x.region_read_off; ra.read_off = x.read_off; ra.read_start = x.read_start; ra.mapq = x.mapq;
			ra.read_skip = x.read_skip; ra.v2_read_bad = x.v2_read_bad; ra.v2_trim_lo = x.v2_trim_lo; ra.v2_trim_hi = x.v2_trim_hi;
			ra.v2_pk = x.v2_pk; ra.v2_hand = x.v2_hand; ra.v2_hoff = x.v2_hoff; ra.n_final = x.n_final; ra.min_overlap_pct = x.min_overlap_pct;
			const bool lpt_on = g_knob.lpt != 0;                   // region order of the combine launch
			ra.lpt_cnt = lpt_on ? wq + 14 * WQ_WORDS : nullptr; ra.lpt_seg = b->lpt_seg.as<int>(); ra.lpt_stride = b->R;
			ra.tier_a_cap = b->v2_arena; ra.tier_b_cap = b->v2_arena_b; ra.n_tier_b = misc + M_NTIERB;
			ra.tier_a_maxc = b->tier_wide ? V3_MAXC : COMB_MAXC_A; ra.manyc_thr = COMB_MAXC_A; ra.n_manyc = misc + M_MANYC;
			for (int k = 0; k < HIST_N; ++k) ra.hist_cap[k] = (int)comb_cap_for(HIST_OCC[k], b->tier_wide ? comb_stat() : comb_stat_a());
			ra.hist = misc + M_HIST;
			ra.min_mapq_assemble = x.min_mapq_assemble; ra.v2_pdw = x.v2_pdw; ra.n_regions = x.n_regions; ra.in_list = x.in_list; ra.n_in = x.n_in;
			ra.out_list = x.out_list; ra.n_out = x.n_out; ra.work_counter = x.work_counter; ra.prof = x.prof; ra.t_start = x.t_start;
			if (b->n_rich) {
				// the read-rich regions (longest chains) on the second stream beside the usual ones
				HIPC(hipEventRecord(b->ev_rfork, s));
				HIPC(hipStreamWaitEvent(s2, b->ev_rfork, 0));
				ReadArgs rl = ra;
				rl.in_list = cl + b->n_small; rl.n_in = cn + 5; rl.v2_pdw = b->v2_pdw_rich; rl.work_counter = wq + 15 * WQ_WORDS;
				hipLaunchKernelGGL((k_asm_reads<8>), dim3(b->grid_v2r_rich), dim3(64), 4 * b->v2_pdw_rich, s2, rl);
				HIPC(hipEventRecord(b->ev_rjoin, s2));
				ra.n_in = cn + 4;                                      // the first n_small entries of the class-1 list
			}
			hipLaunchKernelGGL((k_asm_reads<8>), dim3(b->grid_v2r), dim3(64), 4 * b->v2_pdw, s, ra);
			if (b->n_rich) HIPC(hipStreamWaitEvent(s, b->ev_rjoin, 0));
			x.t_start = nullptr; x.work_counter = wq + 11 * WQ_WORDS; x.v2_pm_dw = b->v2_pm;
			x.lpt_cnt = ra.lpt_cnt; x.lpt_seg = ra.lpt_seg; x.lpt_stride = ra.lpt_stride; x.lpt_nclass = LPT_CLASSES;
			x.out_list = b->retry_listc.as<int>(); x.n_out = misc + M_NRETRYC;
			// The regions by the arena their contigs need (the read phase knows): the first tier on this stream; the second
			// (larger arena, about two thirds of the occupancy) on the second stream beside it; the third (the roomiest, the
			// longest chains) on this stream in front of the first.  A tier the previous batch left empty gets no launch of its
			// own -- an empty launch of workgroups that each ask for 25-60 KB of LDS still has to get every one of them
			// scheduled, 0.05-0.3 ms on the critical path in front of k_ksw while another batch's kernels fill the CUs: the first
			// tier's launch walks its lists behind its own instead (a straggler does not fit there and takes the retry route).
			const bool hint = have_hint;
			// (only behind CLEAN_MIN batches in a row that filed nothing there: see TierHint::clean_c / clean_b.  Until then the second tier keeps a
			// launch of a few workgroups, which takes the odd region where it is filed)
			const bool fold_c = hint && H.n_c == 0 && H.clean_c >= CLEAN_MIN, fold_b = fold_c && H.n_b == 0 && H.clean_b >= CLEAN_MIN;
			HIPC(hipEventRecord(b->ev_bfork, s));
			// The second tier runs beside the first on the other stream -- if its workgroups find LDS: the first tier's persistent
			// grid fills every CU and keeps it until its queue is dry, so a second tier launched next to it in fact ran behind it
			// (C5: 4 % of the regions, the heaviest, one more round of ~1 ms).  When the last batch says how many regions the second
			// tier gets, it is launched with about that many workgroups and the first tier leaves their LDS free on every CU.
			// waves per workgroup: as many as the tier's occupancy leaves wave slots for (32 per CU)
			// (the team build is for 4 waves per SIMD, 128 VGPRs: 16 waves per CU; measured on C5 at 8 workgroups per CU: 2 waves +3 %,
			// 4 waves -30 % -- half the workgroups resident)
			auto launch_comb = [&](int grid, int threads, size_t lds, hipStream_t st, const AsmArgs &args, bool first = false) {
				if (first) {                                           // the build with the short contig table (a region with more contigs takes the retry route)
					if (threads > 64) hipLaunchKernelGGL((k_asm_combine3<4, true, COMB_MAXC_A>), dim3(grid), dim3(threads), lds, st, args);
					else if (g_knob.comb_minw == 7) hipLaunchKernelGGL((k_asm_combine3<7, false, COMB_MAXC_A>), dim3(grid), dim3(64), lds, st, args);
					else if (g_knob.comb_minw == 6) hipLaunchKernelGGL((k_asm_combine3<6, false, COMB_MAXC_A>), dim3(grid), dim3(64), lds, st, args);
					else hipLaunchKernelGGL((k_asm_combine3<5, false, COMB_MAXC_A>), dim3(grid), dim3(64), lds, st, args);
				}
				else if (threads > 64) hipLaunchKernelGGL((k_asm_combine3<4, true>), dim3(grid), dim3(threads), lds, st, args);
				else hipLaunchKernelGGL((k_asm_combine3<5, false>), dim3(grid), dim3(64), lds, st, args);
			};
			auto team = [&](int occ) { const int w = g_knob.comb_waves ? g_knob.comb_waves : occ <= 8 ? 2 : 1; return 64 * std::max(1, std::min(w, (int)V3_MAXW)); };
			const int tm_a = team(b->tier_occ), tm_b = team(std::max(1, (b->tier_occ * 2 + 1) / 3)), tm_c = team(std::max(1, b->tier_occ / 3)), tm_big = team(2);
			int ga = b->grid_v2, gb = b->grid_v2b;
			if (hint && ra.lpt_cnt && !fold_b) {
				const long long nb_hint = H.n_b, reg = std::max(1, H.regions);
				if (nb_hint == 0) gb = std::min(gb, 128);
				else {
					const long long nb = nb_hint * std::max(1, b->n_cls[0]) / reg + 1;
					gb = (int)std::min<long long>(gb, std::max<long long>(g.cus, nb + nb / 4));
					// (never more than a quarter of a CU's LDS: a second tier with thousands of regions is not over in one round, and
					// the first tier must not be left with a workgroup per CU)
					const long long wb_a = comb_wave_bytes(b->v2_arena, b->tier_wide ? comb_stat() : comb_stat_a()), wb_b = comb_wave_bytes(b->v2_arena_b, comb_stat());
					gb = (int)std::min<long long>(gb, g.cus * std::max<long long>(1, g.max_lds / 4 / wb_b));
					const long long per_cu_b = (gb + g.cus - 1) / g.cus;
					const long long occ_a = std::max<long long>(1, ((long long)g.max_lds - per_cu_b * wb_b) / wb_a);
					ga = (int)std::min<long long>(ga, occ_a * g.cus);
				}
			}
			const bool deep_on = ra.lpt_cnt && b->n_deep > 0 && b->grid_v2deep > 0;
			if (deep_on && b->n_cls[0] > b->n_deep) {
				// the wide launch's workgroups need LDS on the CUs the first tier's persistent grid would fill to the last granule (the
				// first tier starts at once, the wide launch behind an event of this stream: it ran BEHIND the first tier -- C3 with
				// 140 regions of 256 reads among 100 000: 5.2 ms for a launch of 2 ms): the first tier leaves them their share
				const long long wb_a = comb_wave_bytes(b->v2_arena, b->tier_wide ? comb_stat() : comb_stat_a());
				const long long wb_d = (g.comb_static_w + 16 + 2ll * b->v2_arena_deep + 4ll * b->v2_pm_deep + LDS_GRAN - 1) / LDS_GRAN * LDS_GRAN;
				const long long per_cu_d = std::min<long long>((b->grid_v2deep + g.cus - 1) / g.cus, 4);
				const long long occ_a = std::max<long long>(1, ((long long)g.max_lds - per_cu_d * wb_d) / wb_a);
				ga = (int)std::min<long long>(ga, occ_a * g.cus);
			}
			if (deep_on) {
				// the regions of more than 255 reads (the longest chains of the batch): the wide build, first on the second stream
				HIPC(hipStreamWaitEvent(s2, b->ev_bfork, 0));
				AsmArgs w = x;
				w.lpt_cnt = ra.lpt_cnt + 3 * LPT_CLASSES; w.lpt_seg = ra.lpt_seg + (size_t)3 * LPT_CLASSES * ra.lpt_stride;
				w.arena_cap = b->v2_arena_deep; w.lds_arena = b->v2_arena_deep; w.v2_pm_dw = b->v2_pm_deep;
				w.work_counter = wq + 24 * WQ_WORDS;
				hipLaunchKernelGGL((k_asm_combine3<4, false, V3_MAXC, true>), dim3(b->grid_v2deep), dim3(64), (size_t)2 * b->v2_arena_deep + 4 * (size_t)b->v2_pm_deep, s2, w);
				HIPC(hipGetLastError());
			}
			if (ra.lpt_cnt) {
				if (!fold_b) {
					if (!deep_on) HIPC(hipStreamWaitEvent(s2, b->ev_bfork, 0));
					AsmArgs y = x;
					y.lpt_cnt = ra.lpt_cnt + 2 * LPT_CLASSES; y.lpt_seg = ra.lpt_seg + (size_t)2 * LPT_CLASSES * ra.lpt_stride;
					y.arena_cap = b->v2_arena_b; y.lds_arena = b->v2_arena_b; y.v2_pm_dw = b->v2_pm_b;
					y.work_counter = wq + 13 * WQ_WORDS;
					launch_comb(gb, tm_b, b->v2_arena_b + 4 * b->v2_pm_b, s2, y);
					HIPC(hipEventRecord(b->ev_bjoin, s2));
				} else HIPC(hipEventRecord(b->ev_bjoin, deep_on ? s2 : s));
				if (!fold_c) {
					AsmArgs z = x;
					z.lpt_cnt = ra.lpt_cnt + LPT_CLASSES; z.lpt_seg = ra.lpt_seg + (size_t)LPT_CLASSES * ra.lpt_stride;
					z.arena_cap = b->v2_arena_c; z.lds_arena = b->v2_arena_c; z.v2_pm_dw = b->v2_pm_c;
					z.work_counter = wq + 16 * WQ_WORDS;
					launch_comb(b->grid_v2c, tm_c, b->v2_arena_c + 4 * b->v2_pm_c, s, z);
				}
				x.lpt_nclass = LPT_CLASSES * (fold_b ? 3 : fold_c ? 2 : 1);
			} else HIPC(hipEventRecord(b->ev_bjoin, s));
			launch_comb(ga, tm_a, b->v2_arena + 4 * b->v2_pm, s, x, !b->tier_wide);
			HIPC(hipStreamWaitEvent(s, b->ev_bjoin, 0));           // (recorded right away when there is no second-tier launch)
			// regions that ran out of room in their launch: the same kernel with a roomy arena, few workgroups per CU -- or, when
			// the previous batch had none, a token launch with the first tier's arena (it gets scheduled at once; a region that
			// does turn up is handed on to the byte-based passes)
			x.in_list = b->retry_listc.as<int>(); x.n_in = misc + M_NRETRYC; x.out_list = b->retry_list0.as<int>(); x.n_out = misc + M_NRETRY0;
			x.lpt_cnt = nullptr;
			x.work_counter = wq + 12 * WQ_WORDS;
			// The retry route -- the roomy launch, then the byte-based passes for what the packed path hands back -- is five
			// launches that are empty for batch after batch, each waiting for wave slots while another batch's persistent grids
			// fill the chip (0.2-0.4 ms in front of k_ksw).  When the last batch needed none of them they are left out, and
			// whoever waits for this run checks the counters: a region that did need the route makes the run repeat in full.
			spec_skip = hint && !g_knob.no_spec && !side && !b->force_full && H.n_big == 0 && H.n_back == 0 && H.clean_r >= CLEAN_MIN;
			if (!spec_skip) {
			if (hint && H.n_big == 0) {
				launch_comb(std::min(b->grid_v2big, 64), tm_a, b->v2_arena + 4 * b->v2_pm, s, x, !b->tier_wide);
			} else {
				x.arena_cap = b->v2_arena_big; x.lds_arena = b->v2_arena_big; x.v2_pm_dw = b->v2_pm_big;
				launch_comb(b->grid_v2big, tm_big, b->v2_arena_big + 4 * b->v2_pm_big, s, x);
			}
			HIPC(hipGetLastError());
			a.arena_seq = nullptr; a.arena_sup = b->lds_sup.as<uint32_t>(); a.arena_cap = b->lds_arena1; a.lds_arena = b->lds_arena1;
			a.in_list = b->retry_list0.as<int>(); a.n_in = misc + M_NRETRY0; a.out_list = o2; a.n_out = misc + M_NRETRY; a.work_counter = wq;
			hipLaunchKernelGGL((k_assemble<64, true, 4>), dim3(b->grid_ovf1), dim3(64), b->lds_arena1, s, a);
			HIPC(hipGetLastError());
			}
		} else if (n1) {
			a.arena_seq = nullptr; a.arena_sup = b->lds_sup.as<uint32_t>(); a.arena_cap = b->lds_arena1; a.lds_arena = b->lds_arena1;
			a.in_list = cl; a.n_in = cn; a.out_list = o2; a.n_out = misc + M_NRETRY; a.work_counter = wq;
			a.t_start = prepacked ? nullptr : tm;                  // the first launch of the stage
			hipLaunchKernelGGL((k_assemble<64, true, 4>), dim3(b->grid_asm), dim3(64), b->lds_arena1, s, a);
			HIPC(hipGetLastError());
		}
		a.t_start = nullptr;
		if (side) HIPC(hipStreamWaitEvent(s, b->ev_join, 0));
		if (!spec_skip) {
			pass2(a, s, o2, misc + M_NRETRY, 4, b->corr.as<Corr>(), b->grid_ovf2);
			pass3(a, s, o3, misc + M_NRETRY2, 5, b->corr.as<Corr>(), b->grid_ovf3);
			pass4(a, s, o4, misc + M_NRETRY3, 6, b->corr.as<Corr>(), b->grid_ovf4);
		}
		HIPC(hipGetLastError());
		spec_skipped_run = spec_skip;
	}
	HIPC(hipEventRecord(b->ev[1], s));
	if (b->R > 0 && b->n_reads > 0) {
		int8_t mat[25];
		ihp_matrix(p.match, p.mismatch, mat);
		KswArgs a;
		a.t_start = tm ? tm + 2 : nullptr;
		a.jobs = b->jobs.as<AlnJob>(); a.n_jobs = misc + M_NJOBS; a.n_jobs_host = 0;
		a.qbase = b->out_seq.as<uint8_t>(); a.tbase = b->ref_bases.as<uint8_t>();
		a.P = make_ksw_params(5, mat, p.gap_open, p.gap_ext, p.bw, p.zdrop, p.ksw_flag, 1);   // ksw2.nim:154-157
		a.lds_budget = b->lds_ksw - 64;
		a.p_scratch = b->p_scratch.as<uint8_t>(); a.p_cap = b->p_cap;
		a.cig_tmp = b->cig_tmp.as<uint32_t>(); a.cig_cap = b->cig_cap;
		a.ez = b->ez.as<KswOut>(); a.cig_off = b->cig_off.as<long long>();
		a.cig_pool = b->cig_pool.as<uint32_t>(); a.cig_cursor = (unsigned long long *)(misc + M_CIG);
		a.cig_pool_cap = b->cig_pool_cap; a.cig_bump_cap = b->cig_bump_cap; a.overflow = misc + M_OVF; a.work_counter = wq + 7 * WQ_WORDS;
		a.prof = profiling ? b->prof.as<long long>() : nullptr;
		a.gm = 0; memset(a.gmat, 0, sizeof(a.gmat));
		g_last_ksw_mode = ksw_mode(a.P);
		a.in_list = nullptr; a.pairs = nullptr; a.ovf_list = b->ksw_ovf.as<int>(); a.ovf_n = misc + M_KSW_OVF;
		if (b->p_cap_pair && ksw_pair_wanted(a.P, g_last_ksw_mode))
		{
			const int prc = launch_ksw_planned(dim3(b->grid_ksw), b->lds_ksw, b->lds_ksw_pair, s, a, b->ksw_plan.as<int>(), (int)std::max<long long>(1, b->njobs_cap),
			                                   wq + 19 * WQ_WORDS, wq + 18 * WQ_WORDS, b->p_scratch_pair.as<uint8_t>(), b->p_cap_pair, b->cig_tmp_pair.as<uint32_t>(),   // (5 sets >= PLAN_ZERO_INTS)
			                                   b->stream2, b->ev_kfork, b->ev_kjoin);
			if (prc) return prc;
		} else launch_ksw(g_last_ksw_mode, dim3(b->grid_ksw), b->lds_ksw, s, a);
		HIPC(hipGetLastError());
		// the roomy launch: left out when the last batch had no such job (a few workgroups that ask for all the LDS of a CU wait for
		// one to drain); the wait checks the count and repeats the run with it otherwise (see the retry launches above)
		ksw_skipped_run = !ksw_roomy;
		if (g_knob.verbose) fprintf(stderr, "[ihp] ksw2: grid %d lds %d p_cap %zu; roomy launch %s: grid %d p_cap %zu cig %d\n", b->grid_ksw, b->lds_ksw, b->p_cap, ksw_skipped_run ? "left out" : "enqueued", b->grid_kovf, b->p_cap_big, b->cig_cap_big);
		if (!ksw_skipped_run) {
			KswArgs r2 = a;
			r2.t_start = nullptr; r2.in_list = b->ksw_ovf.as<int>(); r2.n_jobs = misc + M_KSW_OVF; r2.ovf_list = nullptr; r2.ovf_n = nullptr;
			r2.lds_budget = g.max_lds - 2048 - 64;
			r2.p_scratch = b->p_scratch_big.as<uint8_t>(); r2.p_cap = b->p_cap_big;
			r2.cig_tmp = b->cig_tmp_big.as<uint32_t>(); r2.cig_cap = b->cig_cap_big;
			r2.work_counter = wq + 17 * WQ_WORDS;
			launch_ksw(g_last_ksw_mode, dim3(b->grid_kovf), (size_t)g.max_lds - 2048, s, r2);
			HIPC(hipGetLastError());
		}
	}
	HIPC(hipEventRecord(b->ev[2], s));
	if (b->R > 0 && b->n_reads > 0) {
		TallyArgs a;
		a.t_start = tm ? tm + 4 : nullptr;
		a.jobs = b->jobs.as<AlnJob>(); a.n_jobs = misc + M_NJOBS;
		a.out_seq = b->out_seq.as<uint8_t>(); a.ref_bases = b->ref_bases.as<uint8_t>();
		a.bases = b->bases.as<uint8_t>(); a.mapq = b->mapq.as<uint8_t>();
		{
			const bool have = ((b->v2 && b->n_cls[0] > 0) || b->has_b4 || b->has_pk2) && g_knob.tally_pk;    // k_prepack ran in this chain
			a.pk = have ? b->v2_pk.as<uint32_t>() : nullptr; a.read_bad = have ? b->v2_read_bad.as<uint8_t>() : nullptr;
		}
		a.read_off = b->read_off.as<long long>(); a.region_read_off = b->region_read_off.as<long long>();
		a.ref_origin = b->ref_origin.as<long long>(); a.ctg_start = b->ctg_start.as<long long>();
		a.ez = b->ez.as<KswOut>(); a.cig_off = b->cig_off.as<long long>(); a.cig_pool = b->cig_pool.as<uint32_t>();
		a.P.K = p.K; a.P.min_event_len = p.min_event_len; a.P.max_events = p.max_events; a.P.min_mapq_tally = p.min_mapq_tally;
		a.P.fallback = p.fallback; a.fb_items = p.fallback ? b->fb_items.as<FbItem>() : nullptr; a.fb_count = misc + M_NFB;
		a.hit_pool = b->hit_pool.as<int>(); a.hit_cursor = (unsigned long long *)(misc + M_HIT); a.hit_cap = b->hit_cap;
		a.hit_overflow = misc + M_OVF_HIT; a.hit_region_cnt = b->hitcnt_dev();
		a.hit_bump0 = 2ll * HIT_SLOTS * b->n_reads;
		a.ev_pool = b->ev_pool.as<DevEvent>(); a.ev_cursor = (unsigned long long *)(misc + M_EV);
		a.ev_pool_cap = b->ev_pool_cap; a.ev_off = b->ev_off.as<long long>(); a.n_ev = b->n_ev.as<int>();
		a.overflow = misc + M_OVF; a.work_counter = wq + 8 * WQ_WORDS;
		a.prof = profiling ? b->prof.as<long long>() : nullptr;
		a.lds_bytes = std::min(g.max_lds - 4096, 64 * ((b->max_read_len + 3) / 4 * 4) + 64);
		// with the 2-bit reads at hand a group of 64 reads is a quarter of that; the rare region with a base that is not
		// upper-case ACGT then walks its reads in HBM (tally_reads) instead of staging them
		if (a.pk) a.lds_bytes = std::min(a.lds_bytes, 64 * 4 * ((b->max_read_len + 15) / 16 + 1) + 64);
		a.njobs_cap = (int)std::max<long long>(1, std::min<long long>(b->njobs_cap, 0x7fffffff));
		a.recs = b->tally_recs.as<TallyRec>(); a.rec_cap = b->tally_rec_cap; a.n_recs = misc + M_NRECS; a.ovf_jobs = b->tally_ovf.as<int>();
		hipLaunchKernelGGL(k_tally_prep, dim3((unsigned)((a.njobs_cap + 255) / 256)), dim3(256), 0, s, a);
		if (g_knob.tally_minw == 8) hipLaunchKernelGGL(k_tally<8>, dim3(b->grid_tally), dim3(64), a.lds_bytes, s, a);
		else if (g_knob.tally_minw == 7) hipLaunchKernelGGL(k_tally<7>, dim3(b->grid_tally), dim3(64), a.lds_bytes, s, a);
		else hipLaunchKernelGGL(k_tally<6>, dim3(b->grid_tally), dim3(64), a.lds_bytes, s, a);
		HIPC(hipGetLastError());
	}
	HIPC(hipEventRecord(b->ev[3], s));
	if (b->R > 0 && b->n_reads > 0 && p.fallback) {
		int8_t mat[25];
		ihp_matrix(p.fb_match, p.fb_mismatch, mat);                  // new_ez(mismatch=-2, gap_open=5, gap_ext=1), indelope.nim:318-319
		FbArgs a;
		a.t_start = tm ? tm + 6 : nullptr;
		a.items = b->fb_items.as<FbItem>(); a.n_items = misc + M_NFB; a.max_region_reads = std::max(1, b->max_region_reads);
		a.jobs = b->jobs.as<AlnJob>();
		a.out_seq = b->out_seq.as<uint8_t>(); a.ref_bases = b->ref_bases.as<uint8_t>(); a.bases = b->bases.as<uint8_t>();
		a.quals = b->has_quals ? b->quals.as<uint8_t>() : nullptr; a.mapq = b->mapq.as<uint8_t>();
		a.trim_lo = b->has_trim ? b->trim_lo.as<int>() : nullptr; a.trim_hi = b->has_trim ? b->trim_hi.as<int>() : nullptr;
		a.read_off = b->read_off.as<long long>(); a.region_read_off = b->region_read_off.as<long long>();
		a.read_start = b->read_start.as<long long>(); a.ref_origin = b->ref_origin.as<long long>();
		a.ctg_start = b->ctg_start.as<long long>();
		a.ev_pool = b->ev_pool.as<DevEvent>();
		a.P = make_ksw_params(5, mat, (int8_t)std::abs((int)p.fb_gap_open), (int8_t)std::abs((int)p.fb_gap_ext), p.fb_bw, p.fb_zdrop, p.fb_flag, 1);
		a.min_mapq = p.min_mapq_tally; a.trim_min_qual = p.trim_min_qual;
		a.lds_budget = b->lds_fb - 64;
		a.p_scratch = b->fb_p_scratch.as<uint8_t>(); a.p_cap = b->fb_p_cap;
		a.cig_tmp = b->fb_cig_tmp.as<uint32_t>(); a.cig_cap = b->fb_cig_cap;
		a.overflow = misc + M_OVF; a.work_counter = wq + 9 * WQ_WORDS;
		a.duo = (g_knob.fb_duo ? 1 : 0) | (g_knob.fb_skip ? 2 : 0);
		a.in_list = nullptr; a.n_in = nullptr; a.ovf_list = b->fb_ovf.as<int>(); a.ovf_n = misc + M_FB_OVF; a.ovf_cap = ihp_batch::FB_OVF_CAP;
		hipLaunchKernelGGL(k_fallback, dim3(b->grid_fb), dim3(64), b->lds_fb, s, a);
		HIPC(hipGetLastError());
		// the roomy launch for what that one put on its list: enqueued with the roomy ksw2 launch (same scratch, same streak:
		// the wait checks the list's length and repeats the run in full when this launch was left out and an item needed it)
		const bool fb_roomy = ksw_roomy && b->p_scratch_big.p && b->cig_tmp_big.p;
		fb_roomy_skipped_run = !fb_roomy;
		if (fb_roomy) {
			FbArgs r2 = a;
			r2.t_start = nullptr; r2.in_list = b->fb_ovf.as<int>(); r2.n_in = misc + M_FB_OVF; r2.ovf_list = nullptr; r2.ovf_n = nullptr;
			r2.lds_budget = g.max_lds - 2048 - 64;
			r2.p_scratch = b->p_scratch_big.as<uint8_t>(); r2.p_cap = std::max(b->p_cap_big, b->fb_p_cap_big);
			r2.cig_tmp = b->cig_tmp_big.as<uint32_t>(); r2.cig_cap = std::max(b->cig_cap_big, b->fb_cig_cap_big);
			r2.work_counter = wq + 25 * WQ_WORDS;
			hipLaunchKernelGGL(k_fallback, dim3(std::max(1, b->grid_kovf)), dim3(64), (size_t)g.max_lds - 2048, s, r2);
			HIPC(hipGetLastError());
		}
	}
	HIPC(hipEventRecord(b->ev[5], s));
	if (b->R > 0) {
		SummaryArgs a;
		a.n_regions = b->R; a.region_read_off = b->region_read_off.as<long long>();
		a.status = b->status.as<int>(); a.n_pre = b->n_pre.as<int>(); a.n_final = b->n_final.as<int>();
		a.aln_flags = b->aln_flags.as<int>(); a.n_ev = b->n_ev.as<int>(); a.ev_off = b->ev_off.as<long long>();
		a.ev_pool = b->ev_pool.as<DevEvent>(); a.out = b->summary.as<ihp_region_summary>();
		a.zero = misc; a.n_zero = (int)(b->z_bytes() / sizeof(int)); a.n_report = REPORT_INTS; a.sticky = M_SLAB_BAD;
		HIPC(hipHostGetDevicePointer((void **)&a.report, b->report, 0));
		a.t_end = tm ? (int)(ihp_batch::Z_TIMES / sizeof(int)) + 14 : -1;    // stamp [7]
		hipLaunchKernelGGL(k_summary, dim3((b->R + 255) / 256), dim3(256), 0, s, a);
		HIPC(hipGetLastError());
	}
	HIPC(hipEventRecord(b->ev[4], s));
	b->ran = true;
	b->hint_counted = false;
	b->counted = false;
	if (b->fetch_flags & IHP_FETCH_EAGER) { const int rcp = pack_counts_enqueue(b); if (rcp) return rcp; }
	b->spec_skipped = spec_skipped_run;
	b->ksw_skipped = ksw_skipped_run;
	b->fb_roomy_skipped = fb_roomy_skipped_run;
	b->acc_pending = b->timing;
	b->dirty = false;                                      // k_summary is in the stream: it leaves `misc` clear for the next run
	return 0;
}

static int report_overflow(const ihp_batch *b)
{
	const int *m = b->report;
	if (m[M_SLAB_BAD] || b->slab_bad) {
		const_cast<ihp_batch *>(b)->slab_bad = true;           // (the flag is cleared with the counters; the batch stays refused)
		snprintf(g.err, sizeof(g.err), "compact slab: the read lengths of a region do not add up to its region_base_off step");
		return IHP_E_ARG;
	}
	if (m[M_OVF] || m[M_OVF + 1] || m[M_OVF + 2] || m[M_OVF_HIT]) {
		snprintf(g.err, sizeof(g.err), "device pool overflow: cigar=%d ksw-scratch=%d events=%d hits=%d", m[M_OVF], m[M_OVF + 1],
		         m[M_OVF + 2], m[M_OVF_HIT]);
		return IHP_E_CAPACITY;
	}
	return 0;
}

// Waits for the batch's run.  A run that left out the retry launches (see ihp_batch_run) is repeated in full when a region
// turned out to need them: the counters in the report say so.
static bool spec_failed(const ihp_batch *b)
{
	if (!b->ran) return false;
	if (b->spec_skipped && (b->report[M_NRETRYC] > 0 || b->report[M_NRETRY0] > 0 || g_knob.spec_fail)) return true;
	if (b->fb_roomy_skipped && b->report[M_FB_OVF] > 0) return true;
	return b->ksw_skipped && (b->report[M_KSW_OVF] > 0 || g_knob.spec_fail);
}
static int run_again_in_full(ihp_batch *b)
{
	b->force_full = true;
	const int rc = ihp_batch_run(b);
	b->force_full = false;
	b->n_reruns++;
	return rc;
}
// What the confirmed run needed goes into the hint of the batch's shape (HintTable): every wait that confirms a run calls this.
static void hint_refresh(const ihp_batch *b)
{
	if (!b->ran || b->R <= 0 || !b->hint_key) return;
	TierHint h;
	h.n_b = b->report[M_NTIERB]; h.n_c = b->report[M_NTIERC]; h.n_big = b->report[M_NRETRYC];
	h.n_back = b->report[M_NRETRY0]; h.n_kovf = b->report[M_KSW_OVF] + b->report[M_FB_OVF];   // (what needs the roomy ksw2 / fallback launches)
	// the streaks count RUNS, not waits: sync, fetch, pack and summary all confirm the same run
	const int step = b->hint_counted ? 0 : 1;
	b->hint_counted = true;
	TierHint prev0;
	const bool had0 = g_hints.get(b->hint_key, prev0);
	auto streak = [&](bool ok, int before) { return ok ? std::min(1 << 20, (had0 ? before : 0) + step) : 0; };
	h.clean_r = streak(h.n_big == 0 && h.n_back == 0, prev0.clean_r);
	h.clean_k = streak(h.n_kovf == 0, prev0.clean_k);
	h.clean_c = prev0.clean_c; h.clean_b = prev0.clean_b;
	if (b->v2 && g_knob.lpt) {
		for (int k = 0; k < HIST_N; ++k) h.hist[k] = b->report[M_HIST + k];
		h.n_manyc = b->report[M_MANYC]; h.wide = b->tier_wide ? 1 : 0;
		h.clean_c = streak(h.n_c == 0, prev0.clean_c);
		h.clean_b = streak(h.n_b == 0 && h.n_manyc == 0, prev0.clean_b);
		h.regions = std::max(0, b->n_cls[0] - b->report[M_NRETRY0] - b->n_deep); h.sig = b->tier_sig;   // (the regions the histogram counts: the wide launch's are not filed by arena)
	}
	if (g_knob.verbose) fprintf(stderr, "[ihp] confirmed: %d jobs, %d to the roomy ksw2 launch (skipped %d), overflow flags %d %d %d\n", b->report[M_NJOBS], b->report[M_KSW_OVF], (int)b->ksw_skipped, b->report[M_OVF], b->report[M_OVF + 1], b->report[M_OVF + 2]);
	g_hints.put(b->hint_key, h);
}
static int finish_run(ihp_batch *b)
{
	HIPC(hipStreamSynchronize(b->stream));
	if (b->ran && b->report[M_SLAB_BAD]) b->slab_bad = true;   // (raised once, by k_slab_expand; the counters are cleared with every run)
	if (spec_failed(b)) {
		const int rc = run_again_in_full(b);
		if (rc) return rc;
		HIPC(hipStreamSynchronize(b->stream));
	}
	hint_refresh(b);
	return 0;
}

static void stage_ms_from_report(const ihp_batch *b, float ms[4])
{
	unsigned long long t[8];
	memcpy(t, (const char *)b->report + ihp_batch::Z_TIMES, sizeof(t));
	// stamps: [0] assembly, [2] ksw2, [4] tally, [6] fallback (0 when that stage was not launched), [7] summary
	for (int k = 0; k < 4; ++k) {
		unsigned long long end = 0;
		for (int j = 2 * k + 2; j <= 6 && !end; j += 2) end = t[j];
		if (!end) end = t[7];
		ms[k] = (t[2 * k] && end > t[2 * k] && g.wall_khz > 0) ? (float)((double)(end - t[2 * k]) / (double)g.wall_khz) : 0.0f;
	}
}

// Waits for the batch's run; a pool that overflowed during it (CIGAR / event / hit pools, ksw2 scratch) is reported
// here as IHP_E_CAPACITY, not only when the results are fetched.
extern "C" int ihp_batch_sync(ihp_batch *b)
{
	if (!b) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	{ int rc1 = finish_run(b); if (rc1) return rc1; }
	if (b->ran && b->acc_pending && b->R > 0) {
		// the run's stamps are in the report page (host memory): adding them up here costs no HIP call, so a caller can time
		// every run of a loop without reading anything inside it
		float ms[4];
		stage_ms_from_report(b, ms);
		for (int k = 0; k < 4; ++k) b->acc_ms[k] += ms[k];
		b->acc_n++;
		b->acc_pending = false;
	}
	if (b->ran && b->R > 0) return report_overflow(b);
	return 0;
}

// Mean stage times (as ihp_batch_kernel_ms) over the runs that were waited for with ihp_batch_sync since timing was
// switched on or since the last call with reset != 0.
extern "C" int ihp_batch_kernel_ms_mean(ihp_batch *b, float ms[4], int64_t *n_runs, int reset)
{
	if (!b || !ms) return IHP_E_ARG;
	for (int k = 0; k < 4; ++k) ms[k] = b->acc_n ? (float)(b->acc_ms[k] / (double)b->acc_n) : 0.0f;
	if (n_runs) *n_runs = b->acc_n;
	if (reset) { for (int k = 0; k < 4; ++k) b->acc_ms[k] = 0; b->acc_n = 0; }
	return 0;
}

extern "C" int ihp_batch_stage_ms(ihp_batch *b, float ms[4])
{
	if (!b || !b->ran) return IHP_E_ARG;
	HIPC(hipEventSynchronize(b->ev[4]));
	for (int i = 0; i < 3; ++i) HIPC(hipEventElapsedTime(&ms[i], b->ev[i], b->ev[i + 1]));
	HIPC(hipEventElapsedTime(&ms[3], b->ev[0], b->ev[4]));
	return 0;
}

// Diagnostics (IHP_PROFILE=1): shader-clock cycles summed over waves.
// [0] assemble total, [1] combine, [2] assemble+output, [3] regions; [8] ksw init, [9] ksw DP, [10] ksw traceback, [11] jobs
static int batch_profile64(ihp_batch *b, int64_t out[64]);
// the first `cap` of the 64 counters (a caller built against the 32-entry form of an earlier round passes 32)
extern "C" int ihp_batch_profile_n(ihp_batch *b, int64_t *out, int32_t cap)
{
	if (!b || !out || cap < 0) return IHP_E_ARG;
	int64_t all[64];
	const int rc = batch_profile64(b, all);
	if (rc) return rc;
	memcpy(out, all, sizeof(int64_t) * (size_t)std::min(cap, 64));
	return 0;
}
extern "C" int ihp_batch_profile(ihp_batch *b, int64_t out[64]) { return b && out ? batch_profile64(b, out) : IHP_E_ARG; }
static int batch_profile64(ihp_batch *b, int64_t out[64])
{
	if (!b || !out) return IHP_E_ARG;
	if (!b->ran || !b->work_live) return IHP_E_ARG;           // no run yet, or its scratch went back to the pool (ihp_batch_release_outputs)
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	{ int rc1 = finish_run(b); if (rc1) return rc1; }
	HIPC(hipMemcpy(out, b->prof.p, sizeof(long long) * 64, hipMemcpyDeviceToHost));
	out[24] = b->report[M_NRETRY];                    // regions forwarded at run time to the second pass's overflow list
	out[25] = b->report[M_NRETRY2];                   // ... to the third pass's
	out[26] = b->report[M_NRETRY3];                   // ... and to the catch-all (HBM-arena) pass
	out[22] = g_last_ksw_mode;                        // which k_ksw<MODE> ran
	out[23] = b->report[M_NRETRY0];                   // regions the packed path handed back to the byte-based class-1 kernel
	out[28] = b->report[M_NRETRYC];                   // regions whose contigs needed the roomy combine launch
	out[29] = b->report[M_NTIERB];                    // regions the read phase filed under the second (larger-arena) combine launch
	out[30] = b->report[M_NTIERC];                    // ... and under the third
	out[31] = b->n_reruns;                            // runs of this batch repeated in full because a run without the retry launches met a region that needed them
	out[21] = (b->spec_skipped ? 1 : 0) | (b->ksw_skipped ? 2 : 0) | (b->fb_roomy_skipped ? 4 : 0);   // the last run left out: 1 the assembly retry launches, 2 the roomy ksw2 launch, 4 the fallback's
	out[47] = b->report[M_FB_OVF];                    // items the alignment fallback's main launch handed to its roomy launch
	return 0;
}

extern "C" int ihp_batch_set_timing(ihp_batch *b, int on)
{
	if (!b) return IHP_E_ARG;
	b->timing = on != 0;
	return 0;
}

// Execution time of the stages of the most recent run from device wall-clock stamps: from the moment the stage's first
// workgroup started to the moment the next stage's did (in stream order nothing of a stage runs after that; the last stage
// ends where the summary kernel starts).  ms[0] assemble, [1] ksw2, [2] tally, [3] fallback.
extern "C" int ihp_batch_kernel_ms(ihp_batch *b, float ms[4])
{
	if (!b || !b->ran || !b->timing || !ms) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	{ int rc1 = finish_run(b); if (rc1) return rc1; }
	stage_ms_from_report(b, ms);
	return 0;
}

extern "C" int ihp_batch_fallback_ms(ihp_batch *b, float *ms)
{
	if (!b || !b->ran || !ms) return IHP_E_ARG;
	HIPC(hipEventSynchronize(b->ev[4]));
	HIPC(hipEventElapsedTime(ms, b->ev[3], b->ev[5]));
	return 0;
}

extern "C" int ihp_batch_summary_dev(ihp_batch *b, void **dev_ptr, int64_t *n)
{
	if (!b || !dev_ptr || !n) return IHP_E_ARG;
	// the records of a run are final once the run is confirmed (it may be repeated at the wait): wait here
	if (b->ran && b->work_live) { int rc0 = ensure_init(); if (rc0) return rc0; int rc1 = finish_run(b); if (rc1) return rc1; }
	*dev_ptr = b->summary.p; *n = b->R;
	return 0;
}

// The same address without the wait (it is fixed from upload to free): for a caller that orders its own work behind the run
// by stream or by ihp_batch_sync -- e.g. to enqueue an RCCL gather on another stream -- and must not stall the host here.
// What is behind the pointer is final only after a wait that confirms the run (ihp_batch_sync / fetch / ihp_batch_summary_dev).
extern "C" int ihp_batch_summary_ptr(ihp_batch *b, void **dev_ptr, int64_t *n)
{
	if (!b || !dev_ptr || !n) return IHP_E_ARG;
	*dev_ptr = b->summary.p; *n = b->R;
	return 0;
}

// The per-region summary records of the last run copied to the host (the same records ihp_batch_summary_dev exposes).
extern "C" int ihp_batch_summary_host(ihp_batch *b, ihp_region_summary *out, int64_t cap)
{
	if (!b || !b->ran || (!out && b->R) || cap < b->R) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	{ int rc1 = finish_run(b); if (rc1) return rc1; }
	if (b->R) HIPC(hipMemcpy(out, b->summary.p, sizeof(ihp_region_summary) * (size_t)b->R, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" void ihp_batch_free(ihp_batch *b) { delete b; }

extern "C" int ihp_batch_set_fetch(ihp_batch *b, int32_t flags)
{
	if (!b || (flags & ~(IHP_FETCH_NO_BASES | IHP_FETCH_EAGER | IHP_FETCH_COMPACT))) return IHP_E_ARG;
	b->fetch_flags = flags;
	return 0;
}

// Hands the batch's scratch and result buffers back to the device pool (its inputs and per-region summary records stay):
// a caller that walks through more regions than one GPU holds results for keeps every chunk's inputs resident and only
// one chunk's results at a time.  The results are gone (fetch / pack need another run); the next run takes buffers again.
extern "C" int ihp_batch_release_outputs(ihp_batch *b)
{
	if (!b) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	// the run is confirmed first: one that left launches out and met a region that needed them is repeated here, so that the
	// summary records that stay behind are final (after the release nothing could tell any more)
	int rc = 0;
	if (b->ran && b->work_live) {
		rc = finish_run(b);
		if (!rc && b->R > 0) rc = report_overflow(b);
	} else HIPC(hipStreamSynchronize(b->stream));
	if (b->stream2) HIPC(hipStreamSynchronize(b->stream2));
	release_work(b);
	b->ran = false;
	return rc;
}

// ---- host result slabs ---------------------------------------------------------------
// Every array of an ihp_batch_out lives in ONE pinned host allocation (64-byte header + sections), so that
// the packed device results arrive in a single hipMemcpy at PCIe rate.  Pinning is expensive, so freed slabs
// are kept and reused by later fetches.
namespace {

// section offsets of the flat result arrays inside a slab (the same on the device and on the host)
struct OutLayout {
	size_t status, n_pre, contig_off, ctg_start, ctg_nreads, ctg_seq_off, aln_ref_start, cigar_off, event_off, events,
	       aln_ez, aln_flags, aln_ref_len, ctg_support, cigar, ctg_seq, hit_off, ref_hit, alt_hit, bytes;
	size_t seq4 = 0, sup8 = 0, esc_idx = 0, esc_val = 0; long long esc_cap = 0;       // IHP_FETCH_COMPACT (compact = true): B bases as 4 + 8 bits
	static long long esc_cap_for(long long B) { return B / 128 + 1024; }
	OutLayout(long long R, long long C, long long B, long long W, long long E, long long Hn, bool compact = false) {
		const long long Bc = compact ? B : 0;
		if (compact) B = 0;
		size_t o = 0;
		auto sec = [&](size_t elem, long long n) { const size_t at = o; o += ((size_t)(n > 0 ? n : 1) * elem + 15) / 16 * 16; return at; };
		contig_off = sec(8, R + 1); ctg_start = sec(8, C); ctg_nreads = sec(8, C); ctg_seq_off = sec(8, C + 1);
		aln_ref_start = sec(8, C); cigar_off = sec(8, C + 1); event_off = sec(8, C + 1); hit_off = sec(8, E + 1);
		events = sec(sizeof(ihp_event), E);
		aln_ez = sec(sizeof(ihp_ez), C); status = sec(4, R); n_pre = sec(4, R); aln_flags = sec(4, C); aln_ref_len = sec(4, C);
		ctg_support = sec(4, B); cigar = sec(4, W); ref_hit = sec(4, Hn); alt_hit = sec(4, Hn); ctg_seq = sec(1, B);
		if (compact) {
			esc_cap = esc_cap_for(Bc);
			esc_idx = sec(8, esc_cap + 1); esc_val = sec(4, esc_cap + 1); seq4 = sec(1, (Bc + 1) / 2 + C + 1); sup8 = sec(1, Bc);
		}
		bytes = o;
	}
};
}  // namespace

extern "C" void ihp_free_out(ihp_batch_out *o)
{
	if (!o) return;
	if (o->contig_off) {
		SlabHdr *h = (SlabHdr *)((char *)o->contig_off - sizeof(SlabHdr));   // contig_off is the slab's first section
		if (h->magic == SLAB_MAGIC) g_slabs.put(h);
	}
	memset(o, 0, sizeof(*o));
}

// Carve the arrays of an ihp_batch_out out of a slab laid out by OutLayout and fill in the genotypes
// (indelope.nim:379: genotype(ref_support, alt_support, 1e-3), fp64 on the host).
static void carve_out(char *host, const OutLayout &L, long long R, long long C, long long B, long long W, long long E, long long Hn,
                      double error, ihp_batch_out *out)
{
	out->n_regions = (int32_t)R; out->n_contigs = C; out->n_events = E; out->n_cigar_words = W; out->n_bases = B; out->n_hits = Hn;
	out->ctg_seq4 = out->ctg_sup8 = nullptr; out->n_sup_escapes = 0; out->sup_escape_idx = nullptr; out->sup_escape_val = nullptr;
	out->hit_off = (int64_t *)(host + L.hit_off); out->ref_hit = (int32_t *)(host + L.ref_hit); out->alt_hit = (int32_t *)(host + L.alt_hit);
	out->status = (int32_t *)(host + L.status); out->n_contigs_pre = (int32_t *)(host + L.n_pre);
	out->contig_off = (int64_t *)(host + L.contig_off);
	out->ctg_start = (int64_t *)(host + L.ctg_start); out->ctg_nreads = (int64_t *)(host + L.ctg_nreads);
	out->ctg_seq_off = (int64_t *)(host + L.ctg_seq_off); out->ctg_seq = (uint8_t *)(host + L.ctg_seq);
	out->ctg_support = (uint32_t *)(host + L.ctg_support);
	if (L.esc_cap) {                                                // IHP_FETCH_COMPACT: 4-bit bases, byte supports + escapes (sorted by base index)
		out->ctg_seq = nullptr; out->ctg_support = nullptr;
		out->ctg_seq4 = (uint8_t *)(host + L.seq4); out->ctg_sup8 = (uint8_t *)(host + L.sup8);
		int64_t *ei = (int64_t *)(host + L.esc_idx); uint32_t *ev = (uint32_t *)(host + L.esc_val);
		const long long n = std::min<long long>(ei[0], L.esc_cap);
		std::vector<std::pair<int64_t, uint32_t>> es((size_t)n);
		for (long long k = 0; k < n; ++k) es[(size_t)k] = {ei[1 + k], ev[1 + k]};
		std::sort(es.begin(), es.end());
		for (long long k = 0; k < n; ++k) { ei[1 + k] = es[(size_t)k].first; ev[1 + k] = es[(size_t)k].second; }
		out->n_sup_escapes = n; out->sup_escape_idx = ei + 1; out->sup_escape_val = ev + 1;
	}
	out->aln_flags = (int32_t *)(host + L.aln_flags); out->aln_ref_start = (int64_t *)(host + L.aln_ref_start);
	out->aln_ref_len = (int32_t *)(host + L.aln_ref_len); out->aln_ez = (ihp_ez *)(host + L.aln_ez);
	out->cigar_off = (int64_t *)(host + L.cigar_off); out->cigar = (uint32_t *)(host + L.cigar);
	out->event_off = (int64_t *)(host + L.event_off); out->events = (ihp_event *)(host + L.events);
	for (long long e = 0; e < E; ++e) {
		ihp_event &x = out->events[e];
		if (x.status != IHP_EV_TALLIED) continue;
		ihp_genotype_t gt;
		ihp_genotype(x.ref_support, x.alt_support, error, &gt);
		x.gt = gt.gt; x.gl[0] = gt.gl[0]; x.gl[1] = gt.gl[1]; x.gl[2] = gt.gl[2];
		x.qual = ihp_genotype_qual(&gt);
	}
}

// k_pack_count + k_pack_scan on the batch's stream: per-region counts -> prefix sums; the five totals land in the page-locked
// report block (no copy).  Enqueued by ihp_batch_run itself under IHP_FETCH_EAGER, otherwise by the first pack / fetch.
static int pack_counts_enqueue(ihp_batch *b)
{
	const int R = b->R;
	hipStream_t s = b->stream;
	const size_t S = (size_t)R + 1;
	if (!b->pack_cnt.p) { int rc = b->pack_cnt.alloc(sizeof(long long) * 5 * S); if (rc) return rc; }
	long long *cnt = b->pack_cnt.as<long long>();
	if (R > 0) {
		PackCountArgs a;
		a.R = R; a.region_read_off = b->region_read_off.as<long long>(); a.n_final = b->n_final.as<int>();
		a.ctg_len = b->ctg_len.as<int>(); a.aln_flags = b->aln_flags.as<int>(); a.n_ev = b->n_ev.as<int>();
		a.ez = b->ez.as<KswOut>(); a.ev_off = b->ev_off.as<long long>(); a.ev_pool = b->ev_pool.as<DevEvent>(); a.cnt = cnt;
		hipLaunchKernelGGL(k_pack_count, dim3((R + 255) / 256), dim3(256), 0, s, a);
		HIPC(hipGetLastError());
	}
	hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, R, cnt, (long long *)(b->report + REPORT_INTS));
	HIPC(hipGetLastError());
	b->counted = true;
	return 0;
}

// Results compacted on the device into one slab (k_pack_count -> k_pack_scan -> k_pack); nothing is copied to the host
// except the totals that define the layout.  k_pack is left in flight on the batch's stream (the callers wait).
static int pack_enqueue(ihp_batch *b, void **dev_ptr, int64_t *bytes, int64_t counts[6], bool compact = false)
{
	if (!b->work_live) return IHP_E_ARG;                      // the results went back to the pool (ihp_batch_release_outputs)
	const int R = b->R;
	hipStream_t s = b->stream;
	for (;;) {
		if (!b->counted) { const int rc = pack_counts_enqueue(b); if (rc) return rc; }
		HIPC(hipStreamSynchronize(s));
		if (!spec_failed(b)) break;
		// the run left the retry launches out and a region needed them (ihp_batch_run): again in full, then count again
		const int rc = run_again_in_full(b);
		if (rc) return rc;
	}
	hint_refresh(b);
	const volatile long long *tot_h = (const volatile long long *)(b->report + REPORT_INTS);
	long long tot[5];
	for (int k = 0; k < 5; ++k) tot[k] = tot_h[k];
	long long *cnt = b->pack_cnt.as<long long>();
	if (R > 0) { const int rc = report_overflow(b); if (rc) return rc; }
	// IHP_FETCH_NO_BASES: the contigs' bases and supports stay on the device (ctg_seq_off still tells the lengths)
	const bool no_bases = (b->fetch_flags & IHP_FETCH_NO_BASES) != 0;
	const long long C = tot[0], B = no_bases ? 0 : tot[1], W = tot[2], E = tot[3], Hn = tot[4];
	compact = compact && !no_bases;
	const OutLayout L(R, C, B, W, E, Hn, compact);
	if (b->pack_slab.n < L.bytes) {
		int rc = b->pack_slab.alloc(L.bytes + L.bytes / 8);
		if (rc) return rc;
	}
	char *dev = b->pack_slab.as<char>();
	PackArgs a;
	a.R = R; a.region_read_off = b->region_read_off.as<long long>(); a.ref_origin = b->ref_origin.as<long long>();
	a.status = b->status.as<int>(); a.n_pre = b->n_pre.as<int>(); a.n_final = b->n_final.as<int>();
	a.ctg_len = b->ctg_len.as<int>(); a.aln_flags = b->aln_flags.as<int>(); a.aln_ref_len = b->aln_ref_len.as<int>();
	a.n_ev = b->n_ev.as<int>();
	a.ctg_start = b->ctg_start.as<long long>(); a.ctg_nreads = b->ctg_nreads.as<long long>();
	a.ctg_seq_off = b->ctg_seq_off.as<long long>(); a.aln_ref_start = b->aln_ref_start.as<long long>();
	a.cig_off = b->cig_off.as<long long>(); a.ev_off = b->ev_off.as<long long>();
	a.out_seq = b->out_seq.as<uint8_t>(); a.out_sup = b->out_sup.as<uint32_t>(); a.ez = b->ez.as<KswOut>();
	a.cig_pool = b->cig_pool.as<uint32_t>(); a.ev_pool = b->ev_pool.as<DevEvent>(); a.cnt = cnt;
	a.hit_pool = b->hit_pool.as<int>();
	a.o_hit_off = (int64_t *)(dev + L.hit_off); a.o_ref_hit = (int32_t *)(dev + L.ref_hit); a.o_alt_hit = (int32_t *)(dev + L.alt_hit);
	a.o_status = (int32_t *)(dev + L.status); a.o_n_pre = (int32_t *)(dev + L.n_pre); a.o_contig_off = (int64_t *)(dev + L.contig_off);
	a.o_ctg_start = (int64_t *)(dev + L.ctg_start); a.o_ctg_nreads = (int64_t *)(dev + L.ctg_nreads);
	a.o_ctg_seq_off = (int64_t *)(dev + L.ctg_seq_off);
	a.o_seq = no_bases || compact ? nullptr : (uint8_t *)(dev + L.ctg_seq); a.o_sup = no_bases || compact ? nullptr : (uint32_t *)(dev + L.ctg_support);
	a.o_seq4 = nullptr; a.o_sup8 = nullptr; a.esc_idx = nullptr; a.esc_val = nullptr; a.esc_cap = 0;
	if (compact) {
		a.o_seq4 = (uint8_t *)(dev + L.seq4); a.o_sup8 = (uint8_t *)(dev + L.sup8);
		a.esc_idx = (long long *)(dev + L.esc_idx); a.esc_val = (unsigned *)(dev + L.esc_val); a.esc_cap = L.esc_cap;
		HIPC(hipMemsetAsync(a.esc_idx, 0, 8, s));                   // the escapes' count and the "no 4-bit code" flag
		HIPC(hipMemsetAsync(a.esc_val, 0, 4, s));
	}
	a.o_aln_flags = (int32_t *)(dev + L.aln_flags); a.o_aln_ref_start = (int64_t *)(dev + L.aln_ref_start);
	a.o_aln_ref_len = (int32_t *)(dev + L.aln_ref_len); a.o_ez = (ihp_ez *)(dev + L.aln_ez);
	a.o_cigar_off = (int64_t *)(dev + L.cigar_off); a.o_cigar = (uint32_t *)(dev + L.cigar);
	a.o_event_off = (int64_t *)(dev + L.event_off); a.o_events = (ihp_event *)(dev + L.events);
	hipLaunchKernelGGL(k_pack, dim3(grid_for(R + 1, 16)), dim3(64), 0, s, a);
	HIPC(hipGetLastError());
	*dev_ptr = dev; *bytes = (int64_t)L.bytes;
	counts[0] = R; counts[1] = C; counts[2] = B; counts[3] = W; counts[4] = E; counts[5] = Hn;
	return 0;
}

// The slab stays valid until the batch runs, packs or is freed again.
extern "C" int ihp_batch_pack_dev(ihp_batch *b, void **dev_ptr, int64_t *bytes, int64_t counts[6])
{
	if (!b || !b->ran || !dev_ptr || !bytes || !counts) return IHP_E_ARG;
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	const int rc = pack_enqueue(b, dev_ptr, bytes, counts);
	if (rc) return rc;
	HIPC(hipStreamSynchronize(b->stream));
	return 0;
}

// An ihp_batch_out over a host copy of a packed slab (from ihp_batch_pack_dev on any rank, or ihp_pack_out): the arrays
// point into `slab`, which the caller owns -- do not hand the result to ihp_free_out.
extern "C" int ihp_unpack_slab(void *slab, int64_t bytes, const int64_t counts[6], double error, ihp_batch_out *out)
{
	if (!slab || !counts || !out) return IHP_E_ARG;
	for (int k = 0; k < 6; ++k) if (counts[k] < 0) return IHP_E_ARG;
	const OutLayout L(counts[0], counts[1], counts[2], counts[3], counts[4], counts[5]);
	if ((int64_t)L.bytes > bytes) return IHP_E_CAPACITY;
	memset(out, 0, sizeof(*out));
	carve_out((char *)slab, L, counts[0], counts[1], counts[2], counts[3], counts[4], counts[5], error, out);
	return 0;
}

// Host-side counterpart of the device pack: the arrays of `src` copied into one slab of the same layout (buf may be
// null to ask for the size).  Genotype fields travel as they are.
extern "C" int ihp_pack_out(const ihp_batch_out *src, void *buf, int64_t cap, int64_t *bytes, int64_t counts[6])
{
	if (!src || !bytes || !counts) return IHP_E_ARG;
	const long long R = src->n_regions, C = src->n_contigs, B = src->n_bases, W = src->n_cigar_words, E = src->n_events, Hn = src->n_hits;
	const OutLayout L(R, C, B, W, E, Hn);
	*bytes = (int64_t)L.bytes;
	counts[0] = R; counts[1] = C; counts[2] = B; counts[3] = W; counts[4] = E; counts[5] = Hn;
	if (!buf) return 0;
	if (cap < (int64_t)L.bytes) return IHP_E_CAPACITY;
	char *d = (char *)buf;
	memset(d, 0, L.bytes);
#define CP(off, ptr, n, T) do { if ((n) > 0 && (ptr)) memcpy(d + L.off, (ptr), sizeof(T) * (size_t)(n)); } while (0)
	CP(status, src->status, R, int32_t); CP(n_pre, src->n_contigs_pre, R, int32_t); CP(contig_off, src->contig_off, R + 1, int64_t);
	CP(ctg_start, src->ctg_start, C, int64_t); CP(ctg_nreads, src->ctg_nreads, C, int64_t); CP(ctg_seq_off, src->ctg_seq_off, C + 1, int64_t);
	CP(ctg_seq, src->ctg_seq, B, uint8_t); CP(ctg_support, src->ctg_support, B, uint32_t);
	CP(aln_flags, src->aln_flags, C, int32_t); CP(aln_ref_start, src->aln_ref_start, C, int64_t); CP(aln_ref_len, src->aln_ref_len, C, int32_t);
	CP(aln_ez, src->aln_ez, C, ihp_ez); CP(cigar_off, src->cigar_off, C + 1, int64_t); CP(cigar, src->cigar, W, uint32_t);
	CP(event_off, src->event_off, C + 1, int64_t); CP(events, src->events, E, ihp_event);
	CP(hit_off, src->hit_off, E + 1, int64_t); CP(ref_hit, src->ref_hit, Hn, int32_t); CP(alt_hit, src->alt_hit, Hn, int32_t);
#undef CP
	return 0;
}

extern "C" int ihp_batch_fetch(ihp_batch *b, ihp_batch_out *out)
{
	if (!b || !out || !b->ran) return IHP_E_ARG;
	memset(out, 0, sizeof(*out));
	void *dev = nullptr; int64_t bytes = 0, cnt6[6];
	{ int rc0 = ensure_init(); if (rc0) return rc0; }
	bool compact = (b->fetch_flags & IHP_FETCH_COMPACT) && !(b->fetch_flags & IHP_FETCH_NO_BASES);
	for (;;) {
		int rc = pack_enqueue(b, &dev, &bytes, cnt6, compact);   // k_pack in flight; the copy follows it in stream order: one wait
		if (rc) return rc;
		hipStream_t s = b->stream;
		void *slab = g_slabs.get((size_t)bytes);
		if (!slab) { snprintf(g.err, sizeof(g.err), "hipHostMalloc of %lld bytes failed", (long long)bytes); return IHP_E_NOMEM; }
		char *host = (char *)slab + sizeof(SlabHdr);
		hipError_t e = hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, s);
		if (e == hipSuccess) e = hipStreamSynchronize(s);
		if (e != hipSuccess) { g_slabs.put(slab); return hip_fail(e, "copy of the packed results", __LINE__); }
		const OutLayout L(cnt6[0], cnt6[1], cnt6[2], cnt6[3], cnt6[4], cnt6[5], compact);
		if (compact && (((const uint32_t *)(host + L.esc_val))[0] || ((const int64_t *)(host + L.esc_idx))[0] > L.esc_cap)) {
			// a base outside the 16-letter alphabet, or more supports above 254 than the escape list holds: the plain form
			g_slabs.put(slab);
			compact = false;
			continue;
		}
		carve_out(host, L, cnt6[0], cnt6[1], cnt6[2], cnt6[3], cnt6[4], cnt6[5], b->P.error, out);
		return 0;
	}
}

// Contig c of either form of ihp_batch_out as ASCII bases / 32-bit supports.
extern "C" int ihp_out_contig(const ihp_batch_out *out, int64_t c, uint8_t *seq, uint32_t *sup)
{
	if (!out || c < 0 || c >= out->n_contigs || !out->ctg_seq_off) return IHP_E_ARG;
	const int64_t o0 = out->ctg_seq_off[c], n = out->ctg_seq_off[c + 1] - o0;
	if (out->ctg_seq4) {
		const uint8_t *p4 = out->ctg_seq4 + (o0 >> 1) + c;
		if (seq) for (int64_t i = 0; i < n; ++i) seq[i] = (uint8_t)"=ACMGRSVTWYHKDBN"[(p4[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
		if (sup) {
			for (int64_t i = 0; i < n; ++i) sup[i] = out->ctg_sup8[o0 + i];
			const int64_t *lo = std::lower_bound(out->sup_escape_idx, out->sup_escape_idx + out->n_sup_escapes, o0);
			for (; lo < out->sup_escape_idx + out->n_sup_escapes && *lo < o0 + n; ++lo) sup[*lo - o0] = out->sup_escape_val[lo - out->sup_escape_idx];
		}
		return 0;
	}
	if (!out->ctg_seq || !out->ctg_support) return IHP_E_ARG;                   // IHP_FETCH_NO_BASES: nothing to expand
	if (seq) memcpy(seq, out->ctg_seq + o0, (size_t)n);
	if (sup) memcpy(sup, out->ctg_support + o0, sizeof(uint32_t) * (size_t)n);
	return 0;
}

static void slab_cache_clear() { g_slabs.clear(); }
static void report_pool_clear() { g_reports.clear(); }

extern "C" void *ihp_host_alloc(size_t bytes)
{
	if (ensure_init()) return nullptr;
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
	return p;
}

extern "C" void ihp_host_free(void *p) { if (p) (void)hipHostFree(p); }

// Device -> host copy for callers that hold a device pointer of this library (the slab of ihp_batch_pack_dev, the records
// of ihp_batch_summary_dev) and do not link the HIP runtime themselves.
extern "C" int ihp_copy_to_host(const void *dev_ptr, int64_t bytes, void *out)
{
	if (bytes < 0 || (bytes && (!dev_ptr || !out))) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	if (bytes) HIPC(hipMemcpy(out, dev_ptr, (size_t)bytes, hipMemcpyDeviceToHost));
	return 0;
}

// ---- ROI evidence scan (roi_dev.h) ------------------------------------------------------------------
extern "C" void ihp_free_roi(ihp_roi_out *out)
{
	if (!out) return;
	free(out->roi_start); free(out->roi_stop); free(out->read_off); free(out->reads);
	memset(out, 0, sizeof(*out));
}

extern "C" int ihp_gen_roi(const ihp_roi_in *in, ihp_roi_out *out)
{
	if (!in || !out || in->n_reads < 0 || in->span < 0 || in->max_read_coverage < 0) return IHP_E_ARG;
	if (in->n_reads && (!in->read_start || !in->read_stop || !in->cigar_off || !in->cigar)) return IHP_E_ARG;
	memset(out, 0, sizeof(*out));
	int rc = ensure_init();
	if (rc) return rc;
	hipStream_t s = g.stream;
	const long long N = in->n_reads, len = in->span + 1;
	const long long n_cig = N ? in->cigar_off[N] : 0;
	const long long rblocks = std::max<long long>(1, (N + ROI_BLOCK - 1) / ROI_BLOCK), pblocks = (len + ROI_BLOCK - 1) / ROI_BLOCK;
	// positions relative to the origin
	std::vector<long long> st((size_t)std::max<long long>(N, 1)), en((size_t)std::max<long long>(N, 1));
	for (long long i = 0; i < N; ++i) { st[(size_t)i] = in->read_start[i] - in->origin; en[(size_t)i] = in->read_stop[i] - in->origin; }
	DBuf d_st, d_en, d_skip, d_coff, d_cig, d_pmax, d_bmax, d_ev, d_cut, d_cnt, d_rs, d_re, d_rcnt, d_roff, d_reads;
	if ((rc = d_st.upload(st.data(), sizeof(long long) * (size_t)N, s)) || (rc = d_en.upload(en.data(), sizeof(long long) * (size_t)N, s))) return rc;
	if (in->read_skip && (rc = d_skip.upload(in->read_skip, (size_t)N, s))) return rc;
	static const int64_t zero1[1] = {0};
	if ((rc = d_coff.upload(N ? in->cigar_off : zero1, sizeof(long long) * (size_t)(N + 1), s))) return rc;
	if ((rc = d_cig.upload(in->cigar, sizeof(uint32_t) * (size_t)n_cig, s))) return rc;
	if ((rc = d_pmax.alloc(sizeof(long long) * (size_t)std::max<long long>(N, 1))) || (rc = d_bmax.alloc(sizeof(long long) * (size_t)rblocks))) return rc;
	if ((rc = d_ev.alloc(sizeof(unsigned) * (size_t)len)) || (rc = d_cut.alloc((size_t)len + 2))) return rc;
	if ((rc = d_cnt.alloc(sizeof(long long) * 2 * (size_t)(pblocks + 1)))) return rc;
	if ((rc = d_ev.zero(s)) || (rc = d_cut.zero(s))) return rc;
	RoiArgs a;
	memset(&a, 0, sizeof(a));
	a.n_reads = N; a.len = len; a.start = d_st.as<long long>(); a.stop = d_en.as<long long>();
	a.skip = in->read_skip ? d_skip.as<uint8_t>() : nullptr;
	a.cigar_off = d_coff.as<long long>(); a.cigar = d_cig.as<uint32_t>();
	a.pmax_incl = d_pmax.as<long long>(); a.block_max = d_bmax.as<long long>();
	a.evidence = d_ev.as<unsigned>(); a.cut = d_cut.as<uint8_t>();
	a.min_evidence = in->min_event_support < 0 ? 0 : in->min_event_support > 255 ? 256 : in->min_event_support;
	a.min_reads = in->min_read_coverage; a.max_reads = in->max_read_coverage;
	a.block_cnt = d_cnt.as<long long>(); a.pos_blocks = pblocks;
	if (N > 0) {
		hipLaunchKernelGGL(k_roi_pmax_blocks, dim3((unsigned)rblocks), dim3(ROI_BLOCK), 0, s, a);
		hipLaunchKernelGGL(k_roi_pmax_scan, dim3(1), dim3(ROI_BLOCK), 0, s, a.block_max, rblocks);
		hipLaunchKernelGGL(k_roi_pmax_apply, dim3((unsigned)rblocks), dim3(ROI_BLOCK), 0, s, a);
		hipLaunchKernelGGL(k_roi_evidence, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a);
	}
	hipLaunchKernelGGL(k_roi_count, dim3((unsigned)pblocks), dim3(ROI_BLOCK), 0, s, a);
	hipLaunchKernelGGL(k_roi_count_scan, dim3(1), dim3(ROI_BLOCK), 0, s, a.block_cnt, pblocks);
	HIPC(hipGetLastError());
	long long tot[2] = {0, 0};
	HIPC(hipMemcpyAsync(&tot[0], a.block_cnt + pblocks, sizeof(long long), hipMemcpyDeviceToHost, s));
	HIPC(hipMemcpyAsync(&tot[1], a.block_cnt + 2 * pblocks + 1, sizeof(long long), hipMemcpyDeviceToHost, s));
	HIPC(hipStreamSynchronize(s));
	if (tot[0] != tot[1]) { snprintf(g.err, sizeof(g.err), "roi scan: %lld starts, %lld ends", tot[0], tot[1]); return IHP_E_HIP; }
	const long long R = tot[0];
	std::vector<long long> h_rs((size_t)R), h_re((size_t)R), h_off((size_t)R, -1);
	std::vector<int> h_cnt((size_t)R);
	long long kept = 0, total_idx = 0;
	if (R > 0) {
		if ((rc = d_rs.alloc(sizeof(long long) * (size_t)R)) || (rc = d_re.alloc(sizeof(long long) * (size_t)R)) ||
		    (rc = d_rcnt.alloc(sizeof(int) * (size_t)R)) || (rc = d_roff.alloc(sizeof(long long) * (size_t)R))) return rc;
		a.roi_start = d_rs.as<long long>(); a.roi_end = d_re.as<long long>(); a.n_roi = R; a.roi_cnt = d_rcnt.as<int>();
		hipLaunchKernelGGL(k_roi_emit, dim3((unsigned)pblocks), dim3(ROI_BLOCK), 0, s, a);
		const int grid = grid_for((int)std::min<long long>(R, 1 << 30), 16);
		hipLaunchKernelGGL(k_roi_reads<false>, dim3(grid), dim3(64), 0, s, a);
		HIPC(hipGetLastError());
		HIPC(hipMemcpyAsync(h_rs.data(), a.roi_start, sizeof(long long) * (size_t)R, hipMemcpyDeviceToHost, s));
		HIPC(hipMemcpyAsync(h_re.data(), a.roi_end, sizeof(long long) * (size_t)R, hipMemcpyDeviceToHost, s));
		HIPC(hipMemcpyAsync(h_cnt.data(), a.roi_cnt, sizeof(int) * (size_t)R, hipMemcpyDeviceToHost, s));
		HIPC(hipStreamSynchronize(s));
		for (long long k = 0; k < R; ++k)
			if (h_cnt[(size_t)k] >= in->min_read_coverage && h_cnt[(size_t)k] <= in->max_read_coverage) {   // :485
				h_off[(size_t)k] = total_idx; total_idx += h_cnt[(size_t)k]; kept++;
			}
	}
	out->n_roi = kept; out->n_read_idx = total_idx;
	out->roi_start = (int64_t *)calloc((size_t)std::max<long long>(kept, 1), 8);
	out->roi_stop = (int64_t *)calloc((size_t)std::max<long long>(kept, 1), 8);
	out->read_off = (int64_t *)calloc((size_t)kept + 1, 8);
	out->reads = (int64_t *)calloc((size_t)std::max<long long>(total_idx, 1), 8);
	if (!out->roi_start || !out->roi_stop || !out->read_off || !out->reads) { ihp_free_roi(out); return IHP_E_NOMEM; }
	if (kept > 0) {
		if (total_idx > 0) {
			if ((rc = d_reads.alloc(sizeof(long long) * (size_t)total_idx))) { ihp_free_roi(out); return rc; }
			HIPC(hipMemcpyAsync(d_roff.p, h_off.data(), sizeof(long long) * (size_t)R, hipMemcpyHostToDevice, s));
			a.roi_off = d_roff.as<long long>(); a.roi_reads = d_reads.as<long long>();
			const int grid = grid_for((int)std::min<long long>(R, 1 << 30), 16);
			hipLaunchKernelGGL(k_roi_reads<true>, dim3(grid), dim3(64), 0, s, a);
			HIPC(hipGetLastError());
			HIPC(hipMemcpyAsync(out->reads, d_reads.p, sizeof(long long) * (size_t)total_idx, hipMemcpyDeviceToHost, s));
			HIPC(hipStreamSynchronize(s));
		}
		long long w = 0;
		for (long long k = 0; k < R; ++k) {
			if (h_off[(size_t)k] < 0) continue;
			out->roi_start[w] = h_rs[(size_t)k] + in->origin; out->roi_stop[w] = h_re[(size_t)k] + in->origin;
			out->read_off[w] = h_off[(size_t)k];
			++w;
		}
	}
	out->read_off[kept] = total_idx;
	return 0;
}

// ---- post-tally filters and Variant records (host code, variants_host.h) ------------------
extern "C" int ihp_call_variants(const ihp_params *p, const ihp_batch_in *in, const ihp_batch_out *out, ihp_variants *vars)
{
	if (!p || !in || !out || !vars || p->struct_size != (int32_t)sizeof(ihp_params)) return IHP_E_ARG;
	if (out->n_regions != in->n_regions) return IHP_E_ARG;
	memset(vars, 0, sizeof(*vars));
	return ihp_host::call_variants(*p, *in, *out, vars);
}

extern "C" void ihp_free_variants(ihp_variants *vars)
{
	if (!vars) return;
	free(vars->v); free(vars->chars);
	memset(vars, 0, sizeof(*vars));
}

extern "C" int64_t ihp_format_variant(const ihp_variant *v, const char *chars, const char *chrom, char *buf, int64_t cap)
{
	if (!v || !chars || !chrom) return IHP_E_ARG;
	const std::string line = ihp_host::vcf_line(*v, chars, chrom);
	if (buf && cap > 0) {
		const size_t m = std::min(line.size(), (size_t)cap - 1);
		memcpy(buf, line.data(), m); buf[m] = 0;
	}
	return (int64_t)line.size();
}

extern "C" int ihp_run_regions(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out)
{
	if (!out) return IHP_E_ARG;
	ihp_batch *b = nullptr;
	int rc = ihp_batch_upload(p, in, &b);
	if (rc) return rc;
	rc = ihp_batch_run(b);
	if (!rc) rc = ihp_batch_fetch(b, out);
	ihp_batch_free(b);
	return rc;
}

// ---- multi-GPU: the end-of-job gather over RCCL behind the C ABI (dist_host.h) -----------------------------------
#include "dist_host.h"
