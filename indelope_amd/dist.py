"""Multi-GPU sharding of the per-region path: one process per GPU, contiguous region ranges per rank
(regions are independent), no collective on the data path, and ONE gather of the fixed-size
per-region summary records to rank 0 at the end (RCCL over xGMI with backend "nccl"; "gloo" in CPU tests).
"""
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
