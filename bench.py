#!/usr/bin/env python3
"""bench.py -- candidate regions/sec through assemble + ksw2 + k-mer tally on MI355X.

One "step" = one pass of the whole hot path (three HIP kernels + the per-region summary kernel) over one
batch of synthetic candidate regions that is already resident in HBM.  Default workload = BASELINE.json
configs[1] ("C2": 10k regions x 64 x 150 bp reads, SURVEY.md §8d generator).  With --gpus N each rank
holds its own shard of regions (weak scaling: N x the same per-GPU batch) and every step ends with one
RCCL gather of the fixed-size per-region result records to rank 0.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (HIP events on the library's
stream); `cpu_baseline` times the CPU oracle on this host's cores (checker used as a reported baseline
only -- the Nim reference cannot be built in this image).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
KERNELS = ["k_assemble", "k_ksw", "k_tally"]


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n_int32):
        self.__cuda_array_interface__ = {"shape": (n_int32,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def cpu_baseline(batch, K, want_seconds=12.0):
    """Oracle on the host cores over a bounded sample of the same workload."""
    import oracle
    o = oracle.get()
    used_ref = o.use_reference_ksw(True)
    cores = os.cpu_count() or 1
    p = o.params(K=K)
    probe = batch.slice(0, min(batch.n_regions, 512))
    t0 = time.perf_counter()
    o.bench_regions(probe, p, nthreads=1, reps=1)
    rate1 = probe.n_regions / (time.perf_counter() - t0)
    # all cores: every thread owns a contiguous share of the regions and repeats it `reps` times
    sample = batch.slice(0, min(batch.n_regions, max(cores * 8, 2048)))
    n = sample.n_regions
    reps = max(1, int(rate1 * cores * want_seconds / n / 2))
    t0 = time.perf_counter()
    o.bench_regions(sample, p, nthreads=cores, reps=reps)
    dt = time.perf_counter() - t0
    n = n * reps
    o.use_reference_ksw(False)
    # deterministic work counters of the restatement (secondary rate): per region on the probe sample
    o.run_regions(probe, p)
    cnt = o.counters()
    work = {"char_compares": round(cnt["compares"] / probe.n_regions, 1), "ksw2_dp_cells": round(cnt["dp_cells"] / probe.n_regions, 1),
            "kmer_steps": round(cnt["kmer_steps"] / probe.n_regions, 1)}
    return {"value": round(n / dt, 1), "unit": "regions/s", "cores": cores, "kind": "port",
            "value_1thread": round(rate1, 1), "work_per_region": work,
            "sample": "%d region passes (first regions of the workload, repeated), %d threads over independent regions; C restatement of "
                      "contig.nim/indelope.nim (oracle/), ksw2 = %s; Nim reference not buildable here"
                      % (n, cores, "reference ksw2_extz2_sse.c compiled (oracle/_ref)" if used_ref
                         else "scalar restatement (oracle/_ref absent)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", help="BASELINE config id (C2, C3, C5) for the per-GPU batch")
    ap.add_argument("--regions", type=int, default=0, help="override regions per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--quals", action="store_true",
                    help="hand the base qualities to the device and trim there (indelope.nim:23-38) instead of the "
                         "stager-side trim bounds that SURVEY.md 8b/8d specify as the batch input")
    ap.add_argument("--sub-batches", type=int, default=2,
                    help="the step submits the batch as this many sub-batches of consecutive regions, each an ihp_batch on "
                         "its own stream: while one sub-batch's last regions drain a kernel, the other's next kernel fills "
                         "the chip (+10%% regions/s on C2).  With more than one chain a kernel can wait for wave slots the "
                         "other chain holds, so the per-launch kernel_ms come from device wall-clock stamps "
                         "(ihp_batch_kernel_ms: first workgroup's start -> marker behind the kernel), not from event "
                         "intervals.  1 = one launch chain for the whole batch, timed with HIP events")
    ap.add_argument("--payload", action="store_true",
                    help="with --gpus N > 1: every step also packs the results on the device and sends each rank's slab "
                         "to rank 0 (the variable-length half of the SURVEY 8e gather); off by default, the per-step "
                         "collective is the gather of the fixed-size per-region records")
    ap.add_argument("--dup-frac", type=float, default=0.0,
                    help="fraction of planted events that are tandem duplications (these send the k-mer tally to the "
                         "alignment fallback, indelope.nim:312-372); 0 = the BASELINE workload")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import indelope_amd
    from indelope_amd import synth
    api = indelope_amd.api()
    api.init(local_rank)

    cfg = dict(synth.CONFIGS[args.config])
    R = args.regions or min(cfg["n_regions"], 200_000)
    cfg["n_regions"] = R
    K = cfg["K"]
    batch, _ = synth.generate(first_region=rank * R, dup_frac=args.dup_frac, **cfg)
    if not args.quals:
        batch = batch.with_trim_bounds()                     # A0 on the host (SURVEY 8a row A0, 8b "trim bounds (a,b)")
    params = api.params(K=K)
    S = max(1, min(args.sub_batches, R))
    subs = [batch.slice(R * i // S, R * (i + 1) // S) for i in range(S)]
    hs = [api.batch_upload(sb, params) for sb in subs]
    if S > 1:
        for h in hs:                                         # with several chains in flight a kernel can wait for wave slots:
            api.batch_set_timing(h, True)                    # stage times from device wall-clock stamps, not event intervals
    views = []
    for h in hs:
        sptr, sn = api.batch_summary_dev(h)
        views.append(torch.as_tensor(_DevArray(sptr, sn * 8), device="cuda") if sn else torch.zeros(0, dtype=torch.int32, device="cuda"))
    summary = views[0] if S == 1 else torch.cat(views)       # per-region records of the whole batch, region order
    gather_list = [torch.empty_like(summary) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        for h in hs:
            api.batch_run(h)                                 # asynchronous: the sub-batches' launch chains overlap
        for h in hs:
            api.batch_sync(h)
        if world > 1:
            if S > 1:
                torch.cat(views, out=summary)
            dist.gather(summary, gather_list, dst=0)
            if args.payload:
                from indelope_amd import dist as idist
                for h in hs:
                    ptr, nbytes, counts = api.batch_pack_dev(h)
                    slab = torch.as_tensor(_DevBytes(ptr, nbytes), device="cuda")
                    idist.gather_payload(slab, counts, rank, world, dst=0)

    for _ in range(args.warmup):
        step()
    stage = np.zeros(4)
    fb_ms = 0.0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for h in hs:                                         # per launch: the mean over the sub-batches' launches
            ev = np.array(api.batch_stage_ms(h))
            if S > 1:
                km = api.batch_kernel_ms(h)
                ev[:3] = km[:3]
                fb_ms += km[3] / S
            else:
                fb_ms += api.batch_fallback_ms(h) / S
            stage += ev / S
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage /= max(args.steps, 1)
    fb_ms /= max(args.steps, 1)

    if rank == 0:
        from indelope_amd.host import concat_results
        parts = [api.batch_fetch(h) for h in hs]
        res = concat_results(parts)
        assert (res.status == 0).all(), "regions failed on the device"
        # B = sum_reads(len+9) + len_refwindow + sum_contigs(5 len+16) + sum_aln(44+4 n_cigar) + sum_events(2K+12)
        alg_bytes = batch.algorithmic_input_bytes() + res.algorithmic_output_bytes(K)
        # the same terms split by the kernel that moves them, per LAUNCH: a launch processes one sub-batch
        by_kernel = {k: 0 for k in KERNELS}
        for sb, pr in zip(subs, parts):
            for k, v in pr.algorithmic_bytes_by_kernel(sb, K).items():
                by_kernel[k] += v / S
        by_kernel = {k: int(v) for k, v in by_kernel.items()}
        dom = int(np.argmax(stage[:3]))
        achieved = by_kernel[KERNELS[dom]] / (stage[dom] * 1e-3) / 1e9
        traffic = None          # HBM bytes per launch of the dominant kernel from the committed PMC passes (same workload only)
        pmc = os.path.join(ROOT, "profiles", "r01_c2_pmc.json")
        if args.config == "C2" and R == 10_000 and S == 2 and os.path.exists(pmc):
            k = json.load(open(pmc))["kernels"]
            # the stage is one launch of each of these (the later assembly passes are empty on this workload)
            names = {"k_assemble": ("k_assemble<64, true, 4>",), "k_ksw": ("k_ksw<3>", "k_ksw<4>"), "k_tally": ("k_tally",)}[KERNELS[dom]]
            t = [k[n]["traffic"] for n in names if n in k]
            traffic = int(sum(t)) if t else None
        out = {
            "metric": "candidate regions/sec (assemble+ksw2+kmer-genotype), 150bp x 64-read batches",
            "value": round(world * R * args.steps / dt, 1), "unit": "regions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d regions/GPU x %s reads x %d bp, K=%d, err %g (SURVEY 8d generator, seed 0x1DE10BE^%d)"
                                   % (args.config, R, "%d-%d" % cfg["n_reads"] if cfg["n_reads"][0] != cfg["n_reads"][1]
                                      else str(cfg["n_reads"][0]), cfg["read_len"], K, cfg["err_rate"], cfg["config_id"])
                                   + (", %g of events tandem duplications" % args.dup_frac if args.dup_frac else ""),
                       "read_trim": "device, from base qualities" if args.quals else "stager (trim bounds in the batch)",
                       "submission": "%d sub-batch%s of consecutive regions per step, each on its own stream" % (S, "es" if S > 1 else ""),
                       "regions_per_gpu": R, "sharding": "contiguous region ranges per rank, one RCCL gather of "
                       "per-region result records per step" if world > 1 else "single GPU"},
            "kernel_ms": dict({k: round(float(v), 4) for k, v in zip(KERNELS + ["total"], stage)}, k_fallback=round(fb_ms, 4)),
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(by_kernel[KERNELS[dom]]),
                         "algorithmic_bytes_per_region": round(by_kernel[KERNELS[dom]] * S / R, 1),
                         "launches_per_step": S,
                         "algorithmic_bytes_by_kernel": by_kernel,
                         "whole_path": {"algorithmic_bytes_per_step": int(alg_bytes),
                                        "achieved": round(alg_bytes / (dt / args.steps) / 1e9, 2),
                                        "frac": round(alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}},
            "results": {"contigs": int(res.n_contigs), "events": int(res.n_events),
                        "tallied": int((res.events["status"] == 0).sum()),
                        "fallback_events": int((res.events["aligned"] == 1).sum())},
        }
        if os.environ.get("IHP_PROFILE"):
            out["profile_cycles"] = [int(x) for x in sum(np.array(api.batch_profile(h)) for h in hs)]
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch, K)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            # the work-based rate beside the HBM roofline: what the reference algorithm would have executed per second
            w = out["cpu_baseline"]["work_per_region"]
            out["work_rate"] = {"char_compares_per_s": round(w["char_compares"] * out["value"], -6),
                                "ksw2_dp_cells_per_s": round(w["ksw2_dp_cells"] * out["value"], -6),
                                "kmer_steps_per_s": round(w["kmer_steps"] * out["value"], -6)}
        print(json.dumps(out))
    for h in hs:
        api.batch_free(h)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
