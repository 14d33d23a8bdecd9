"""Multi-GPU sharding of the per-region path: one process per GPU, contiguous region ranges per rank
(regions are independent), no collective on the data path, and ONE gather of the fixed-size
per-region summary records to rank 0 at the end.

On GPUs the gather is the library's own (`Communicator`: ctypes over ihp_dist_* of include/indelope_hip.h, which
drives librccl directly -- the same entry points the Nim host binds, nim/indelope_hip.nim); torch is then only
the launcher (torchrun's RANK / WORLD_SIZE and the store that carries the 128-byte id).  The torch functions
below (`gather_summaries`, `gather_payload`) are the same exchange on a torch process group: what the CPU tests
run on gloo, where there is no device and no RCCL.
"""
import ctypes as C

import numpy as np

from . import _abi as A

SUMMARY_WORDS = A.SUMMARY_DTYPE.itemsize // 4      # int32 words per ihp_region_summary
PAD_STATUS = -999


def shard_bounds(weights, world):
    """Contiguous shard boundaries [world+1] balancing sum(weights) (e.g. reads per region); keeps region order."""
    w = np.asarray(weights, np.float64)
    n = len(w)
    if n == 0:
        return np.zeros(world + 1, np.int64)
    c = np.concatenate([[0.0], np.cumsum(w)])
    targets = c[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(c, targets, side="left")
    b = np.concatenate([[0], np.clip(cuts, 0, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def shard_batch(batch, rank, world, by_reads=True):
    """The sub-batch of `rank` (RegionBatch.slice keeps the arrays contiguous)."""
    per_region = np.diff(batch.region_read_off)
    w = per_region.astype(np.float64) * (per_region + 8) if by_reads else np.ones(batch.n_regions)
    b = shard_bounds(w, world)
    return batch.slice(int(b[rank]), int(b[rank + 1])), b


def summaries_from_result(res):
    """Host-side construction of the same records the k_summary kernel leaves on the device."""
    out = np.zeros(res.n_regions, A.SUMMARY_DTYPE)
    out["status"], out["n_contigs_pre"] = res.status, res.n_contigs_pre
    out["n_contigs"] = np.diff(res.contig_off)
    out["ref_support"] = out["alt_support"] = -1
    for r in range(res.n_regions):
        first = True
        for c in range(res.contig_off[r], res.contig_off[r + 1]):
            if not res.aln_flags[c] & A.IHP_ALN_DONE:
                continue
            out["n_aligned"][r] += 1
            ev = res.events[res.event_off[c]:res.event_off[c + 1]]
            out["n_events"][r] += len(ev)
            for e in ev:
                if e["status"] == A.IHP_EV_TALLIED:
                    if first:
                        out["ref_support"][r], out["alt_support"][r] = e["ref_support"], e["alt_support"]
                        first = False
                    out["n_tallied"][r] += 1
    return out


def gather_payload(slab, counts, rank, world, dst=0, force=False):
    """The variable-length half of the end-of-job gather: every rank's packed result slab (a uint8
    tensor on the process group's device: the device slab of ihp_batch_pack_dev under "nccl", host bytes under "gloo")
    travels to `dst` point to point -- each peer over its own xGMI link to the root, no ring -- after one all_gather of
    the seven int64 that size it.  Returns [(slab tensor, counts)] in rank order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    meta = torch.tensor([slab.numel()] + [int(c) for c in counts], dtype=torch.int64, device=slab.device)
    if world == 1 and not force:                           # force: the collective runs even in a group of one (tests)
        return [(slab, meta[1:].cpu().numpy())]
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    if rank != dst:
        if slab.numel():
            dist.send(slab, dst=dst)
        return None
    out, pending = [], []
    for r in range(world):                                 # all receives are posted before any is waited for: the
        n = int(metas[r][0].item())                        # peers transmit at the same time, each on its own link
        if r == dst:
            buf = slab
        else:
            buf = torch.empty(n, dtype=torch.uint8, device=slab.device)
            if n:
                pending.append(dist.irecv(buf, src=r))
        out.append((buf, metas[r][1:].cpu().numpy()))
    for q in pending:
        q.wait()
    return out


def gather_summaries(local, rank, world, dst=0, force=False):
    """One gather of the per-region records to `dst`.  `local`: int32 tensor [n_local, SUMMARY_WORDS] on the
    device the process group uses.  Shards may differ in size: they are padded to the longest and trimmed
    again on `dst`.  Returns the concatenation in rank (= region) order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.full((m, SUMMARY_WORDS), PAD_STATUS, dtype=torch.int32, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


class Communicator:
    """ihp_dist over ctypes: ncclCommInitRank on the library's device, grouped send / recv to the root (dist_host.h).

    `api` is the HipApi of this process (api.init(local_gpu) done); `exchange_id(id_bytes_or_None) -> bytes` carries
    rank 0's 128-byte id to every rank (a torch.distributed broadcast, a file, MPI ...): it gets the id on rank 0 and
    None elsewhere, and returns the id everywhere."""

    def __init__(self, api, rank, world, exchange_id):
        self.api, self.rank, self.world = api, int(rank), int(world)
        buf = (C.c_uint8 * A.IHP_DIST_ID_BYTES)()
        mine = None
        if self.rank == 0:
            api._chk_hip(api.b.dist_unique_id(buf, A.IHP_DIST_ID_BYTES), "dist_unique_id")
            mine = bytes(buf)
        ident = exchange_id(mine)
        if len(ident) != A.IHP_DIST_ID_BYTES:
            raise ValueError("the id is %d bytes, not %d" % (len(ident), A.IHP_DIST_ID_BYTES))
        C.memmove(buf, ident, A.IHP_DIST_ID_BYTES)
        self.h = C.c_void_p()
        api._chk_hip(api.b.dist_init(self.rank, self.world, buf, A.IHP_DIST_ID_BYTES, C.byref(self.h)), "dist_init")

    def _out(self, total):
        """A page-locked buffer for `total` records on the root, kept between calls (a pageable one costs the copy 5x)."""
        if getattr(self, "_pin_n", 0) < total:
            if getattr(self, "_pin", None):
                self.api.b.host_free(self._pin)
            self._pin_n = max(1, int(total) + int(total) // 4)
            self._pin = self.api.b.host_alloc(self._pin_n * A.SUMMARY_DTYPE.itemsize)
            if not self._pin:
                raise MemoryError("ihp_host_alloc")
        return np.ctypeslib.as_array(C.cast(self._pin, C.POINTER(C.c_uint8)), (self._pin_n * A.SUMMARY_DTYPE.itemsize,)).view(A.SUMMARY_DTYPE)

    def _counts(self, counts):
        if counts is None:
            return None, None
        c = np.ascontiguousarray(counts, np.int64)
        assert len(c) == self.world
        return c, A.ptr(c, A.i64p)

    def gather_records(self, dev_ptr, n, root=0, counts=None, cap=None):
        """n 32-byte records at device address dev_ptr (final) -> SUMMARY_DTYPE array in rank order on `root`, None elsewhere.
        The array is a view of the communicator's page-locked buffer: valid until the next gather."""
        c, cp = self._counts(counts)
        total = int(c.sum()) if c is not None else (int(cap) if cap is not None else None)
        if total is None:
            raise ValueError("counts or cap: the root has to size its buffer")
        out = self._out(total) if self.rank == root else None
        n_total, counts_out = C.c_int64(), np.zeros(self.world, np.int64)
        self.api._chk_hip(self.api.b.dist_gather_records(self.h, C.c_void_p(dev_ptr), int(n), root, cp,
                                                         out.ctypes.data_as(C.c_void_p) if out is not None else None, total,
                                                         C.byref(n_total), A.ptr(counts_out, A.i64p)), "dist_gather_records")
        return (out[:n_total.value], counts_out) if self.rank == root else None

    def gather_summaries(self, batch_handle, n_regions, root=0, counts=None, cap=None):
        """The records of a batch (its run is waited for and confirmed first)."""
        c, cp = self._counts(counts)
        total = int(c.sum()) if c is not None else (int(cap) if cap is not None else int(n_regions) * self.world)
        out = self._out(total) if self.rank == root else None
        n_total, counts_out = C.c_int64(), np.zeros(self.world, np.int64)
        self.api._chk_hip(self.api.b.dist_gather_summaries(self.h, batch_handle, root, cp,
                                                           out.ctypes.data_as(C.c_void_p) if out is not None else None, total,
                                                           C.byref(n_total), A.ptr(counts_out, A.i64p)), "dist_gather_summaries")
        return (out[:n_total.value], counts_out) if self.rank == root else None

    def gather_payload(self, batch_handle, root=0):
        """Every rank's full results on `root`: [BatchResult] in rank (= region) order and the bytes received per rank."""
        from .host import BatchResult
        outs = (A.BatchOut * self.world)()
        nbytes = np.zeros(self.world, np.int64)
        self.api._chk_hip(self.api.b.dist_gather_payload(self.h, batch_handle, root, outs, A.ptr(nbytes, A.i64p)), "dist_gather_payload")
        if self.rank != root:
            return None
        res = []
        try:
            for r in range(self.world):
                res.append(BatchResult(outs[r]))
        finally:
            for r in range(self.world):
                self.api.b.free_out(C.byref(outs[r]))
        return res, nbytes

    def close(self):
        if getattr(self, "_pin", None):
            self.api.b.host_free(self._pin)
            self._pin, self._pin_n = None, 0
        if self.h:
            self.api.b.dist_finalize(self.h)
            self.h = C.c_void_p()


def torch_id_exchange(device=None):
    """exchange_id for Communicator over an initialised torch.distributed group (any backend): one 128-byte broadcast."""
    import torch
    import torch.distributed as dist

    def ex(mine):
        t = torch.zeros(A.IHP_DIST_ID_BYTES, dtype=torch.uint8, device=device if device is not None else "cpu")
        if mine is not None:
            t.copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return ex


def file_id_exchange(path, timeout=120.0):
    """exchange_id through a file every rank can see (no torch at all: what a Nim / C launcher would do)."""
    import os
    import time

    def ex(mine):
        if mine is not None:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(mine)
            os.replace(tmp, path)
            return mine
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                raise TimeoutError("no id at " + path)
            time.sleep(0.01)
        return open(path, "rb").read()
    return ex
