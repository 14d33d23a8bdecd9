// dist_host.h -- the ONE collective of the path behind the C ABI (include/indelope_hip.h, "multi-GPU"): every rank's per-region
// records (and, on request, its packed results) to the root, in rank = region order, which is what the reference's main loop
// needs for its sequential last-two-variants dedupe (indelope.nim:601-608).  Included at the end of indelope_hip.hip (it uses
// the library's context, slab cache and batch internals).
//
// librccl is NOT linked: librccl.so.1 is opened on the first ihp_dist_* call, so a single-GPU caller never pays for loading it
// and this library loads on a box without it.  xGMI is point to point: peers send straight to the root (grouped ncclSend /
// ncclRecv -- seven links into the root carry seven peers at once); there is no ring and no reduction.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>                     // types and prototypes only: every call goes through the table below

namespace {

struct RcclApi {
	void *lib = nullptr;
	bool tried = false;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
	std::lock_guard<std::mutex> lk(g_rccl_mu);
	if (g_rccl.lib) return 0;
	if (g_rccl.tried) { snprintf(g.err, sizeof(g.err), "librccl.so.1 could not be opened (tried before)"); return IHP_E_UNSUPPORTED; }
	g_rccl.tried = true;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *h = nullptr;
	for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!h) { snprintf(g.err, sizeof(g.err), "dlopen(librccl.so.1): %s", dlerror()); return IHP_E_UNSUPPORTED; }
#define RS(field, sym) do { g_rccl.field = (decltype(g_rccl.field))dlsym(h, sym); if (!g_rccl.field) { snprintf(g.err, sizeof(g.err), "librccl: no symbol %s", sym); dlclose(h); return IHP_E_UNSUPPORTED; } } while (0)
	RS(GetUniqueId, "ncclGetUniqueId"); RS(CommInitRank, "ncclCommInitRank"); RS(CommDestroy, "ncclCommDestroy");
	RS(GroupStart, "ncclGroupStart"); RS(GroupEnd, "ncclGroupEnd"); RS(Send, "ncclSend"); RS(Recv, "ncclRecv");
	RS(AllGather, "ncclAllGather"); RS(GetErrorString, "ncclGetErrorString");
#undef RS
	g_rccl.lib = h;
	return 0;
}

int rccl_fail(ncclResult_t e, const char *what, int line)
{
	snprintf(g.err, sizeof(g.err), "rccl: %s (line %d): %s", what, line, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
	return IHP_E_HIP;
}
#define NCCLC(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) return rccl_fail(e_, #x, __LINE__); } while (0)

static_assert(sizeof(ncclUniqueId) == IHP_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
static_assert(sizeof(ihp_region_summary) == 32, "ihp_region_summary is 32 bytes");

}  // namespace

struct ihp_dist {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1;
	hipStream_t stream = nullptr;
	DBuf meta;                                  // [world][8] int64: what the sizing all-gather lands in (the rank's own row is its send buffer)
	DBuf recv;                                  // root: the records of every rank, rank order
	std::vector<DBuf> slabs;                    // root: one packed result slab per peer (gather_payload)
	long long *meta_host = nullptr;             // page-locked mirror of `meta`
};

extern "C" int ihp_dist_unique_id(void *id, int64_t cap)
{
	if (!id || cap < IHP_DIST_ID_BYTES) return IHP_E_ARG;
	int rc = rccl_load();
	if (rc) return rc;
	ncclUniqueId u;
	NCCLC(g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return 0;
}

extern "C" int ihp_dist_init(int32_t rank, int32_t world, const void *id, int64_t id_bytes, ihp_dist **out)
{
	if (!out || !id || id_bytes < IHP_DIST_ID_BYTES || world < 1 || rank < 0 || rank >= world) return IHP_E_ARG;
	*out = nullptr;
	int rc = ensure_init();
	if (rc) return rc;
	if ((rc = rccl_load())) return rc;
	ihp_dist *d = new (std::nothrow) ihp_dist();
	if (!d) return IHP_E_NOMEM;
	d->rank = rank; d->world = world;
	ncclUniqueId u;
	memcpy(&u, id, sizeof(u));
	{
		const ncclResult_t e = g_rccl.CommInitRank(&d->comm, world, u, rank);
		if (e != ncclSuccess) { delete d; return rccl_fail(e, "ncclCommInitRank", __LINE__); }
	}
	hipError_t he = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
	if (he == hipSuccess) he = hipHostMalloc((void **)&d->meta_host, sizeof(long long) * 8 * (size_t)world, hipHostMallocDefault);
	if (he != hipSuccess || (rc = d->meta.alloc(sizeof(long long) * 8 * (size_t)world))) {
		const int r2 = he != hipSuccess ? hip_fail(he, "ihp_dist_init", __LINE__) : rc;
		(void)ihp_dist_finalize(d);
		return r2;
	}
	*out = d;
	return 0;
}

extern "C" int ihp_dist_rank(const ihp_dist *d) { return d ? d->rank : IHP_E_ARG; }
extern "C" int ihp_dist_world(const ihp_dist *d) { return d ? d->world : IHP_E_ARG; }

extern "C" int ihp_dist_finalize(ihp_dist *d)
{
	if (!d) return 0;
	int rc = 0;
	if (d->stream) (void)hipStreamSynchronize(d->stream);
	if (d->comm && g_rccl.CommDestroy) { const ncclResult_t e = g_rccl.CommDestroy(d->comm); if (e != ncclSuccess) rc = rccl_fail(e, "ncclCommDestroy", __LINE__); }
	if (d->stream) (void)hipStreamDestroy(d->stream);
	if (d->meta_host) (void)hipHostFree(d->meta_host);
	delete d;
	return rc;
}

// Every rank's row of `nvals` int64 (<= 8) to every rank: d->meta_host[r * 8 + k] on return.
static int dist_exchange_meta(ihp_dist *d, const long long *mine, int nvals)
{
	long long *dev = d->meta.as<long long>();
	long long row[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int k = 0; k < nvals; ++k) row[k] = mine[k];
	HIPC(hipMemcpyAsync(dev + 8 * d->rank, row, sizeof(row), hipMemcpyHostToDevice, d->stream));
	HIPC(hipStreamSynchronize(d->stream));                      // (`row` is a local)
	NCCLC(g_rccl.AllGather(dev + 8 * d->rank, dev, 8, ncclInt64, d->comm, d->stream));   // in place: the rank's row sits at its own offset
	HIPC(hipMemcpyAsync(d->meta_host, dev, sizeof(long long) * 8 * (size_t)d->world, hipMemcpyDeviceToHost, d->stream));
	HIPC(hipStreamSynchronize(d->stream));
	return 0;
}

extern "C" int ihp_dist_gather_records(ihp_dist *d, const void *dev_records, int64_t n, int32_t root, const int64_t *counts_in,
                                       ihp_region_summary *out, int64_t cap, int64_t *n_total, int64_t *counts_out)
{
	if (!d || n < 0 || (n && !dev_records) || root < 0 || root >= d->world) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	const int W = d->world;
	std::vector<long long> cnt((size_t)W);
	if (counts_in) {
		for (int r = 0; r < W; ++r) { if (counts_in[r] < 0) return IHP_E_ARG; cnt[(size_t)r] = counts_in[r]; }
		if (cnt[(size_t)d->rank] != n) return IHP_E_ARG;
	} else {
		const long long mine = n;
		if ((rc = dist_exchange_meta(d, &mine, 1))) return rc;
		for (int r = 0; r < W; ++r) cnt[(size_t)r] = d->meta_host[8 * r];
	}
	const size_t REC = sizeof(ihp_region_summary);
	if (d->rank != root) {
		if (n) {
			NCCLC(g_rccl.Send(dev_records, (size_t)n * REC, ncclUint8, root, d->comm, d->stream));
			HIPC(hipStreamSynchronize(d->stream));                  // the caller may run the batch again as soon as this returns
		}
		return 0;
	}
	long long total = 0;
	for (int r = 0; r < W; ++r) total += cnt[(size_t)r];
	if (n_total) *n_total = total;
	if (counts_out) for (int r = 0; r < W; ++r) counts_out[r] = cnt[(size_t)r];
	// (a short buffer: the peers' sends are matched all the same -- the exchange stays collective and the communicator usable)
	const bool fits = out && cap >= total;
	if (d->recv.n < (size_t)total * REC) { if ((rc = d->recv.alloc((size_t)std::max<long long>(1, total + total / 4) * REC))) return rc; }
	char *dst = d->recv.as<char>();
	NCCLC(g_rccl.GroupStart());                                     // every receive posted before any is waited for
	long long off = 0;
	ncclResult_t ge = ncclSuccess;
	for (int r = 0; r < W; ++r) {
		if (r != root && cnt[(size_t)r] && ge == ncclSuccess) ge = g_rccl.Recv(dst + (size_t)off * REC, (size_t)cnt[(size_t)r] * REC, ncclUint8, r, d->comm, d->stream);
		off += cnt[(size_t)r];
	}
	{ const ncclResult_t e2 = g_rccl.GroupEnd(); if (ge == ncclSuccess) ge = e2; }
	if (ge != ncclSuccess) return rccl_fail(ge, "ncclRecv (grouped)", __LINE__);
	off = 0;
	for (int r = 0; r < root; ++r) off += cnt[(size_t)r];
	if (n) HIPC(hipMemcpyAsync(dst + (size_t)off * REC, dev_records, (size_t)n * REC, hipMemcpyDeviceToDevice, d->stream));
	if (fits && total) HIPC(hipMemcpyAsync(out, dst, (size_t)total * REC, hipMemcpyDeviceToHost, d->stream));
	HIPC(hipStreamSynchronize(d->stream));
	return fits || total == 0 ? 0 : IHP_E_CAPACITY;
}

extern "C" int ihp_dist_gather_summaries(ihp_dist *d, ihp_batch *b, int32_t root, const int64_t *counts_in,
                                         ihp_region_summary *out, int64_t cap, int64_t *n_total, int64_t *counts_out)
{
	if (!d || !b || !b->ran) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	// the records are final once the run is confirmed (a run that left launches out may be repeated here)
	if (b->work_live) { if ((rc = finish_run(b))) return rc; if (b->R > 0 && (rc = report_overflow(b))) return rc; }
	else HIPC(hipStreamSynchronize(b->stream));
	return ihp_dist_gather_records(d, b->summary.p, b->R, root, counts_in, out, cap, n_total, counts_out);
}

extern "C" int ihp_dist_gather_payload(ihp_dist *d, ihp_batch *b, int32_t root, ihp_batch_out *outs, int64_t *bytes_out)
{
	if (!d || !b || !b->ran || root < 0 || root >= d->world || (d->rank == root && !outs)) return IHP_E_ARG;
	int rc = ensure_init();
	if (rc) return rc;
	const int W = d->world;
	void *dev = nullptr; int64_t bytes = 0, c6[6] = {0, 0, 0, 0, 0, 0};
	// a rank whose pack fails still takes part in the sizing exchange (bytes = -1), so that nobody waits for a slab that never comes
	const int prc = ihp_batch_pack_dev(b, &dev, &bytes, c6);
	long long mine[7] = {prc ? -1 : (long long)bytes, c6[0], c6[1], c6[2], c6[3], c6[4], c6[5]};
	if ((rc = dist_exchange_meta(d, mine, 7))) return rc;
	for (int r = 0; r < W; ++r) if (d->meta_host[8 * r] < 0) {
		if (!prc) snprintf(g.err, sizeof(g.err), "ihp_dist_gather_payload: rank %d could not pack its results", r);
		return prc ? prc : IHP_E_HIP;
	}
	if (d->rank != root) {
		if (bytes) { NCCLC(g_rccl.Send(dev, (size_t)bytes, ncclUint8, root, d->comm, d->stream)); HIPC(hipStreamSynchronize(d->stream)); }
		return 0;
	}
	for (int r = 0; r < W; ++r) memset(&outs[r], 0, sizeof(ihp_batch_out));
	if ((int)d->slabs.size() < W) d->slabs = std::vector<DBuf>((size_t)W);
	for (int r = 0; r < W; ++r) {
		const size_t nb = (size_t)d->meta_host[8 * r];
		if (r != root && d->slabs[(size_t)r].n < nb && (rc = d->slabs[(size_t)r].alloc(nb + nb / 8))) return rc;
	}
	NCCLC(g_rccl.GroupStart());
	ncclResult_t ge = ncclSuccess;
	for (int r = 0; r < W; ++r) {
		const size_t nb = (size_t)d->meta_host[8 * r];
		if (r != root && nb && ge == ncclSuccess) ge = g_rccl.Recv(d->slabs[(size_t)r].p, nb, ncclUint8, r, d->comm, d->stream);
	}
	{ const ncclResult_t e2 = g_rccl.GroupEnd(); if (ge == ncclSuccess) ge = e2; }
	if (ge != ncclSuccess) return rccl_fail(ge, "ncclRecv (grouped)", __LINE__);
	// device slabs -> page-locked host slabs of the result cache (what ihp_free_out hands back), all copies in flight together
	std::vector<void *> host((size_t)W, nullptr);
	auto undo = [&]() { for (void *h : host) if (h) g_slabs.put(h); for (int r = 0; r < W; ++r) memset(&outs[r], 0, sizeof(ihp_batch_out)); };
	for (int r = 0; r < W; ++r) {
		const size_t nb = (size_t)d->meta_host[8 * r];
		host[(size_t)r] = g_slabs.get(std::max<size_t>(nb, 64));
		if (!host[(size_t)r]) { undo(); snprintf(g.err, sizeof(g.err), "hipHostMalloc of %zu bytes failed", nb); return IHP_E_NOMEM; }
		const void *src = r == root ? dev : d->slabs[(size_t)r].p;
		if (nb) { const hipError_t e = hipMemcpyAsync((char *)host[(size_t)r] + sizeof(SlabHdr), src, nb, hipMemcpyDeviceToHost, d->stream); if (e != hipSuccess) { undo(); return hip_fail(e, "copy of a gathered slab", __LINE__); } }
		if (bytes_out) bytes_out[r] = (int64_t)nb;
	}
	{ const hipError_t e = hipStreamSynchronize(d->stream); if (e != hipSuccess) { undo(); return hip_fail(e, "ihp_dist_gather_payload", __LINE__); } }
	for (int r = 0; r < W; ++r) {
		const long long *m = d->meta_host + 8 * r;
		const OutLayout L(m[1], m[2], m[3], m[4], m[5], m[6]);
		if ((long long)L.bytes > m[0]) { undo(); snprintf(g.err, sizeof(g.err), "ihp_dist_gather_payload: rank %d sent %lld bytes for a layout of %zu", r, m[0], L.bytes); return IHP_E_ARG; }
		carve_out((char *)host[(size_t)r] + sizeof(SlabHdr), L, m[1], m[2], m[3], m[4], m[5], m[6], b->P.error, &outs[r]);
	}
	return 0;
}
