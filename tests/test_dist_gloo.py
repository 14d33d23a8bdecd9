"""world_size-2 test of the sharding + gather path on CPU (gloo).  The per-rank compute is the CPU oracle here
(no GPU in this container); on the GPU box the same code runs with backend "nccl" and the HIP library."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from indelope_amd import _abi as A
from indelope_amd import dist as idist
from indelope_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_regions, outfile):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    o = oracle.get()
    batch, _ = synth.generate(n_regions, n_reads=(8, 48), err_rate=1e-3, config_id=31)
    shard, bounds = idist.shard_batch(batch, rank, world)
    res = o.run_regions(shard)
    local = torch.from_numpy(idist.summaries_from_result(res).view(np.int32).reshape(-1, idist.SUMMARY_WORDS).copy())
    allsum = idist.gather_summaries(local, rank, world, dst=0)
    if rank == 0:
        np.save(outfile, allsum.numpy())
    else:
        assert allsum is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    n_regions, world = 37, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_regions, out), nprocs=world, join=True)
    got = np.load(out)
    batch, _ = synth.generate(n_regions, n_reads=(8, 48), err_rate=1e-3, config_id=31)
    exp = idist.summaries_from_result(oracle.run_regions(batch)).view(np.int32).reshape(-1, idist.SUMMARY_WORDS)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)          # sharding changes nothing; rank order == region order


def test_shard_bounds_partition():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 100):
            w = rng.integers(1, 300, n)
            b = idist.shard_bounds(w, world)
            assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all() and len(b) == world + 1
    b = idist.shard_bounds(np.ones(80), 8)
    assert np.diff(b).tolist() == [10] * 8


def test_batch_slices_are_self_contained(oracle):
    batch, _ = synth.generate(20, n_reads=(8, 32), err_rate=1e-3, config_id=32)
    whole = oracle.run_regions(batch)
    parts = [oracle.run_regions(batch.slice(lo, hi)) for lo, hi in ((0, 7), (7, 13), (13, 20))]
    assert sum(p.n_contigs for p in parts) == whole.n_contigs
    assert np.array_equal(np.concatenate([p.ctg_seq for p in parts]), whole.ctg_seq)
    assert np.array_equal(np.concatenate([p.events["ref_support"] for p in parts]), whole.events["ref_support"])


def _payload_worker(rank, world, port, n_regions, outfile):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import indelope_amd
    import oracle
    from indelope_amd.host import concat_results
    o = oracle.get()
    api = indelope_amd.api()                           # host-side pack / unpack of the product library (no GPU needed)
    batch, _ = synth.generate(n_regions, n_reads=(8, 48), err_rate=1e-3, config_id=34, dup_frac=0.2)
    shard, _ = idist.shard_batch(batch, rank, world)
    res = o.run_regions(shard)
    slab, counts = api.pack_out(res)                   # on a GPU rank: the device slab of ihp_batch_pack_dev
    got = idist.gather_payload(torch.from_numpy(slab.copy()), counts, rank, world, dst=0)
    if rank == 0:
        parts = [api.unpack_slab(s.numpy(), c) for s, c in got]
        whole = concat_results(parts)
        np.savez(outfile, **{f: getattr(whole, f) for f in whole.FIELDS})
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_payload_gather(tmp_path, oracle):
    """The variable-length results of every rank, packed into slabs, gathered to rank 0 and unpacked, equal the results
    of the whole batch (headers first, then the payload)."""
    from indelope_amd.host import BatchResult
    n_regions, world = 41, 3
    out = str(tmp_path / "payload.npz")
    mp.spawn(_payload_worker, args=(world, _free_port(), n_regions, out), nprocs=world, join=True)
    z = np.load(out)
    batch, _ = synth.generate(n_regions, n_reads=(8, 48), err_rate=1e-3, config_id=34, dup_frac=0.2)
    exp = oracle.run_regions(batch)
    for f in BatchResult.FIELDS:
        a, b = z[f], getattr(exp, f)
        assert a.shape == b.shape, f
        if f == "events":
            for n in b.dtype.names:
                assert np.array_equal(a[n], b[n]) or n in ("gl", "qual"), n
            assert np.allclose(a["gl"], b["gl"]) and np.allclose(a["qual"], b["qual"])
        else:
            assert np.array_equal(a, b), f


def test_pack_unpack_round_trip(oracle):
    import indelope_amd
    from indelope_amd.host import BatchResult
    api = indelope_amd.api()
    for n in (0, 1, 23):
        batch, _ = synth.generate(max(n, 1), n_reads=(8, 40), err_rate=1e-3, config_id=35, dup_frac=0.3)
        if n == 0:
            batch = batch.slice(0, 0)
        res = oracle.run_regions(batch)
        slab, counts = api.pack_out(res)
        back = api.unpack_slab(slab, counts)
        assert BatchResult.first_difference(back, res) is None
        assert counts.tolist() == [res.n_regions, res.n_contigs, len(res.ctg_seq), len(res.cigar), res.n_events, len(res.ref_hit)]
