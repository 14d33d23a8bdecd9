#!/usr/bin/env python3
"""bench.py -- candidate regions/sec through assemble + ksw2 + k-mer tally on MI355X.

One "step" = one pass of the whole hot path (the HIP kernels + the per-region summary kernel) over one batch of
synthetic candidate regions that is already resident in HBM.  Default workload = BASELINE.json configs[1] ("C2":
10k regions x 64 x 150 bp reads, SURVEY.md 8d generator).

  python bench.py                       one GPU, C2
  python bench.py --gpus N              starts N ranks itself (one process per GPU over RCCL; the parent never touches
                                        the GPU) -- or run it under `python -m torch.distributed.run --nproc-per-node N`
  --scaling weak   (default)            every rank holds `--in-flight` (2) resident C2-sized batches of its own that take turns:
                                        a step is one pass over one of them, and a batch is run again as soon as its own last run
                                        is done, so two steps are in flight (the path's callers stream independent batches; rounds
                                        1-3 ran ONE batch as two sub-batches and waited for every step: `--in-flight 1 --lockstep`,
                                        also measured in `other_configs`); a timed block ends with ONE RCCL gather of the fixed-size
                                        per-region records to rank 0
  --scaling strong --config C4          the 5 M regions of BASELINE configs[3] split over the ranks by
                                        indelope_amd.dist.shard_bounds, each rank generating its shard from the seed
                                        (first_region), walked in resident chunks; one gather per step; `--payload`
                                        adds the variable-length result slabs

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel; `cpu_baseline` times the CPU oracle on
this host's cores (the checker, used as a reported baseline only -- the Nim reference cannot be built in this image);
`e2e` is the PCIe-inclusive rate (host buffers in -> host buffers out), never `value`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hardware queues for the streams of the batches in flight (read when the HIP runtime starts: before torch or the library
# make their first call; see ihp_init in indelope_hip.hip); never overrides the caller's setting
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ISSUE_PEAK = 1.0               # wave64 instructions per cycle per CU and per unit: a 16-lane SIMD takes 4 cycles per VALU instruction and
                               # the scalar unit serves one SIMD per cycle, so a CU retires at most 1 VALU + 1 SALU per cycle (rocprofv3's
                               # VALUBusy / SALUBusy definitions; tools/ubench_issue.hip reaches 0.93 / 0.96 by wall clock)
KERNELS = ["k_assemble", "k_ksw", "k_tally"]
ISSUE_PEAK_ARCH = 2.0          # MI355X_MICROARCH.md: a SIMD-32 issues a wave64 VALU instruction over 2 cycles -- reached only by pure streams of
                               # mov / and / xor / add (1.5-1.8 measured); in a mixed stream every VALU instruction costs the half-rate slot
                               # (tools/ubench_ksw.hip: v_sub / v_max alternating 0.93 per cycle and CU)
PMC_FILE = os.path.join("profiles", "r06_c2_pmc.json")
MIX_FILE = os.path.join("profiles", "r06_c2_pmc_mix.json")
STAGE_MEMBERS = {"k_assemble": ("k_prepack", "k_prepack_fast", "k_slab_expand", "k_asm_reads", "k_asm_combine3", "k_assemble"),
                 "k_ksw": ("k_ksw", "k_ksw_pair", "k_ksw_plan_count", "k_ksw_plan_place"), "k_tally": ("k_tally_prep", "k_tally")}


def src_sha16():
    """Hash of the library's sources: the committed PMC passes carry it, so that a bench line can say whether the counters it
    quotes were taken from the code it just ran."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "indelope_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip", ".cpp")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n_int32):
        self.__cuda_array_interface__ = {"shape": (n_int32,), "typestr": "<i4", "data": (ptr, False), "version": 2}


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


# ----------------------------------------------------------------------------------------------- CPU baseline
def _host_cpus():
    """CPUs this process may actually use: affinity mask, capped by a cgroup CPU quota if there is one."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    info = {"os_cpu_count": os.cpu_count(), "affinity": aff, "cgroup_quota_cpus": quota}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=5).stdout
        for key, name in (("Model name", "model"), ("Socket(s)", "sockets"), ("Core(s) per socket", "cores_per_socket"),
                          ("Thread(s) per core", "threads_per_core")):
            for ln in txt.splitlines():
                if ln.strip().startswith(key + ":"):
                    v = ln.split(":", 1)[1].strip()
                    info[name] = int(v) if v.isdigit() else v
                    break
    except Exception:
        pass
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return usable, info


def cpu_baseline(batch, K, want_seconds=16.0):
    """Oracle on the host cores over a bounded sample of the same workload: a thread sweep 1, 2, 4, ... up to the CPUs
    this process owns (affinity / cgroup quota, not os.cpu_count()); `value` is the best point of the curve."""
    import oracle
    o = oracle.get()
    used_ref = o.use_reference_ksw(True)
    usable, info = _host_cpus()
    p = o.params(K=K)
    probe = batch.slice(0, min(batch.n_regions, 512))
    t0 = time.perf_counter()
    o.bench_regions(probe, p, nthreads=1, reps=1)
    rate1 = probe.n_regions / (time.perf_counter() - t0)
    counts = sorted({1, usable} | {c for c in (2, 4, 8, 16, 32, 64, 128, 256) if c < usable}
                    | ({info.get("cores_per_socket")} if isinstance(info.get("cores_per_socket"), int)
                       and info["cores_per_socket"] <= usable else set()))
    per_point = want_seconds / len(counts)
    curve = []
    for c in counts:
        # every thread owns a contiguous share of the sample and repeats it `reps` times
        n = min(batch.n_regions, max(c * 16, 1024))
        sample = batch.slice(0, n)
        reps = max(1, int(rate1 * c * 0.8 * per_point / n))
        t0 = time.perf_counter()
        o.bench_regions(sample, p, nthreads=c, reps=reps)
        dt = time.perf_counter() - t0
        curve.append({"threads": c, "regions_per_s": round(n * reps / dt, 1), "region_passes": n * reps})
    o.use_reference_ksw(False)
    best = max(curve, key=lambda x: x["regions_per_s"])
    # deterministic work counters of the restatement (secondary rate): per region on the probe sample
    o.run_regions(probe, p)
    cnt = o.counters()
    work = {"char_compares": round(cnt["compares"] / probe.n_regions, 1), "ksw2_dp_cells": round(cnt["dp_cells"] / probe.n_regions, 1),
            "kmer_steps": round(cnt["kmer_steps"] / probe.n_regions, 1)}
    one = curve[0]["regions_per_s"]
    out = {"value": best["regions_per_s"], "unit": "regions/s", "cores": best["threads"], "kind": "port",
           "value_1thread": one, "curve": curve, "host": info, "cpus_usable": usable,
           "scaling_vs_1thread": round(best["regions_per_s"] / one, 2),
           "work_per_region": work,
           "sample": "first regions of the workload, %d region passes at the best point (%d threads over independent regions, each "
                     "repeating its share); C restatement of contig.nim/indelope.nim (oracle/), ksw2 = %s; Nim reference not "
                     "buildable here" % (best["region_passes"], best["threads"],
                                         "reference ksw2_extz2_sse.c compiled (oracle/_ref)" if used_ref
                                         else "scalar restatement (oracle/_ref absent)")}
    cps = info.get("cores_per_socket")
    if isinstance(cps, int):
        # the north-star's "single-socket CPU": the curve point at one socket's cores if this process owns that many,
        # else the linear extrapolation from the per-thread rate at the best point (stated as such)
        pt = [c for c in curve if c["threads"] == cps]
        out["single_socket"] = ({"cores": cps, "regions_per_s": pt[0]["regions_per_s"], "how": "measured"} if pt else
                                {"cores": cps, "regions_per_s": round(best["regions_per_s"] / best["threads"] * cps, 1),
                                 "how": "extrapolated linearly from %d threads (this process owns %d CPUs)" % (best["threads"], usable)})
    return out


# ------------------------------------------------------------------------------------------------------ e2e
def e2e_rates(api, batch, params, threads=3, reps=5, depth=2):
    """Host buffers in -> host buffers out through the C ABI, PCIe included (never `value`).
    `slab`: what a stager written for this library hands over -- ONE page-locked slab per batch (ihp_slab_layout: 4-bit
    bases as BAM stores them, trim bounds), uploaded with a single copy; results fetched without the contigs' bases and
    supports (IHP_FETCH_NO_BASES: events, k-mer counts, CIGARs, alignment records) or in full.  `threads` host threads
    each keep `depth` batches of their own in flight (ihp_batch_upload_slab and ihp_batch_run only enqueue: a thread
    starts batch k+1 before it waits for and fetches batch k), so uploads, runs and fetches of different batches overlap.
    `arrays`: the separate pageable ASCII arrays of ihp_batch_in through ihp_run_regions (round 2's leg)."""
    import ctypes as C
    import threading
    from indelope_amd import _abi as A

    def start(slab, no_bases):                               # no_bases: (IHP_FETCH_NO_BASES, IHP_FETCH_COMPACT)
        h = api.batch_upload_slab2(slab, params)
        api.batch_set_fetch(h, no_bases=no_bases[0], compact=no_bases[1], eager=True)
        api.batch_run(h)
        return h

    def finish(h, no_bases):
        api.batch_sync(h)
        t2 = time.perf_counter()
        out = A.BatchOut()
        rc = api.b.batch_fetch(h, C.byref(out))              # device pack + one copy + genotype(): the C call alone
        assert rc == 0, rc
        t3 = time.perf_counter()
        api.b.free_out(C.byref(out))
        api.batch_free(h)
        return t2, t3

    def one(slab, no_bases):
        t0 = time.perf_counter()
        h = api.batch_upload_slab2(slab, params)
        api.batch_sync(h)                                    # the copy alone (the call itself only enqueues it)
        t1 = time.perf_counter()
        api.batch_set_fetch(h, no_bases=no_bases[0], compact=no_bases[1], eager=True)
        api.batch_run(h)
        t2, t3 = finish(h, no_bases)
        return (t1 - t0, t2 - t1, t3 - t2, t3 - t0)

    slabs = [[api.make_slab2(batch) for _ in range(max(depth, 2))] for _ in range(threads)]
    # one batch at a time, handed over as two halves: the second half's copy runs beside the first half's kernels, and two launch
    # chains fill each other's tails (the batch is still alone on the device from its first byte to its last result)
    half = batch.n_regions // 2
    halves = [api.make_slab2(batch.slice(0, half)), api.make_slab2(batch.slice(half, batch.n_regions))] if half > 0 else None

    def one_split(no_bases):
        t0 = time.perf_counter()
        hs_ = []
        for sl in halves:
            h = api.batch_upload_slab2(sl, params)
            api.batch_set_fetch(h, no_bases=no_bases[0], compact=no_bases[1], eager=True)
            api.batch_run(h)
            hs_.append(h)
        for h in hs_:
            finish(h, no_bases)
        return time.perf_counter() - t0
    try:
        res = {}
        # full: every contig's bases and supports come back too -- 4 bits + a byte per base (IHP_FETCH_COMPACT, round 6; the host
        # expands what it reads: ihp_out_contig); full_plain: the same as ASCII bytes + 32-bit supports (rounds 3-5)
        for name, nb in (("events_only", (True, False)), ("full", (False, True)), ("full_plain", (False, False))):
            t = np.array([one(slabs[0][0], nb) for _ in range(reps + 1)][1:]) * 1e3
            med = np.median(t, axis=0)
            split_ms = float(np.median([one_split(nb) for _ in range(reps + 1)][1:])) * 1e3 if halves else None
            n_each = max(16, reps * 6)

            gate = threading.Barrier(threads + 1)

            def worker(k, nb=nb):
                warm = [start(slabs[k][j], nb) for j in range(depth)]    # untimed, with as many batches in flight as the timed loop keeps:
                for h in warm:                                           # this thread's buffers come out of the pools from here on
                    finish(h, nb)
                gate.wait()
                q = []
                for i in range(n_each):
                    q.append(start(slabs[k][i % depth], nb))         # a slab is reused only after its batch has been fetched
                    if len(q) == depth:
                        finish(q.pop(0), nb)
                while q:
                    finish(q.pop(0), nb)
            # three times, the median reported (and all three): one run in five of this leg has a thread waiting out a page-locked
            # allocation or a queue it shares (1.6-2.6 M regions/s against 5.2-5.4)
            dts = []
            for _rep in range(3):
                gate = threading.Barrier(threads + 1)
                th = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
                for x in th:
                    x.start()
                gate.wait()
                t0 = time.perf_counter()
                for x in th:
                    x.join()
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[1]
            # ONE host thread with three batches in flight on three streams (upload and run only enqueue; the thread waits for and
            # fetches the oldest batch while the copies and kernels of the two behind it run)
            pool = [sl for row in slabs for sl in row][:4]
            d1 = 3
            q = [start(pool[j], nb) for j in range(d1)]
            for h in q:
                finish(h, nb)
            q, n1 = [], n_each * 2
            host_up = 0.0
            t0 = time.perf_counter()
            for i in range(n1):
                tu = time.perf_counter()
                q.append(start(pool[i % len(pool)], nb))         # (four slabs for three in flight: a slab is reused after its batch's fetch)
                host_up += time.perf_counter() - tu
                if len(q) == d1:
                    finish(q.pop(0), nb)
            while q:
                finish(q.pop(0), nb)
            dt1 = time.perf_counter() - t0
            res[name] = {"one_batch_ms": {k: round(float(v), 3) for k, v in zip(("upload", "run", "fetch", "total"), med)},
                         "one_batch_regions_per_s": round(batch.n_regions / (med[3] * 1e-3), 1),
                         "one_batch_as_two_halves": ({"ms": round(split_ms, 3), "regions_per_s": round(batch.n_regions / (split_ms * 1e-3), 1)} if split_ms else None),
                         "sustained": {"threads": threads, "in_flight_per_thread": depth, "batches": n_each * threads,
                                       "ms_per_batch": round(dt / (n_each * threads) * 1e3, 3),
                                       "ms_per_batch_of_three_runs": [round(x / (n_each * threads) * 1e3, 3) for x in dts],
                                       "regions_per_s": round(batch.n_regions * n_each * threads / dt, 1)},
                         "sustained_one_thread": {"threads": 1, "in_flight": d1, "batches": n1, "ms_per_batch": round(dt1 / n1 * 1e3, 3),
                                                  "regions_per_s": round(batch.n_regions * n1 / dt1, 1),
                                                  "host_ms_in_upload_and_run_calls": round(host_up / n1 * 1e3, 3)}}
        slab_bytes = int(slabs[0][0].layout.bytes)
    finally:
        for row in slabs:
            for sl in row:
                sl.free()
        for sl in halves or []:
            sl.free()
    # round 2's leg for comparison: separate pageable arrays, ASCII bases, full results
    n_each = max(3, reps)

    def worker_arrays_once():
        out = A.BatchOut()
        cin = batch.as_c()
        assert api.b.run_regions(C.byref(params), C.byref(cin), C.byref(out)) == 0
        api.b.free_out(C.byref(out))

    def worker_arrays():
        for _ in range(n_each):
            worker_arrays_once()
    gate = threading.Barrier(threads + 1)

    def timed_arrays():
        worker_arrays_once()
        gate.wait()
        worker_arrays()
    th = [threading.Thread(target=timed_arrays) for _ in range(threads)]
    for x in th:
        x.start()
    gate.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    ev = res["events_only"]
    return {"one_batch_ms": ev["one_batch_ms"], "one_batch_regions_per_s": ev["one_batch_regions_per_s"], "sustained": ev["sustained"],
            "sustained_one_thread": ev["sustained_one_thread"], "one_batch_as_two_halves": ev["one_batch_as_two_halves"],
            "slab_MB": round(slab_bytes / 1e6, 2),
            "inputs": "one page-locked COMPACT slab per batch (ihp_slab2_layout, %d bytes: 4-bit read bases, 14 bytes per read, 2-bit windows), "
                      "a single upload copy, the arrays made on the device; results without contig bases/supports (IHP_FETCH_NO_BASES)" % slab_bytes,
            "full_results": dict(res["full"], form="IHP_FETCH_COMPACT: contig bases 4 bits each, supports a byte each + escapes (1.5 bytes per base instead of 5)",
                                 plain_form={k: res["full_plain"][k] for k in ("one_batch_ms", "one_batch_regions_per_s", "sustained", "sustained_one_thread")}),
            "pageable_arrays": {"threads": threads, "batches": n_each * threads, "ms_per_batch": round(dt / (n_each * threads) * 1e3, 3),
                                "regions_per_s": round(batch.n_regions * n_each * threads / dt, 1),
                                "inputs": "ihp_run_regions on separate pageable arrays, ASCII bases, full results"},
            "note": "upload + run + device pack + fetch + genotype through the C ABI; PCIe-inclusive, never `value`"}


# ------------------------------------------------------------------------------------------ other configs
def quick_config(api, name, regions, steps, warmup, check=True, sub_batches=1, dup_frac=0.0, in_flight=2, lockstep=False, check_regions=20_000):
    """A short run of another BASELINE config on this GPU, submitted like the headline (`in_flight` resident batches of `regions`
    regions take turns, a step is one pass over one of them as `sub_batches` launch chains; `lockstep`: every step waited for):
    value, per-launch stage times, the HBM-roofline fraction of its dominant stage, oracle check."""
    from indelope_amd import synth
    from indelope_amd.host import BatchResult, concat_results
    cfg = dict(synth.CONFIGS[name])
    K = cfg["K"]
    params = api.params(K=K)
    cfg["n_regions"] = regions
    B, S = max(1, in_flight), max(1, sub_batches)
    cuts = [b * regions + regions * i // S for b in range(B) for i in range(S)] + [B * regions]
    subs, hs = [], []
    for i in range(B * S):
        g = dict(cfg)
        g["n_regions"] = cuts[i + 1] - cuts[i]
        sb, _ = synth.generate(first_region=cuts[i], dup_frac=dup_frac, **g)
        sb = sb.with_trim_bounds()
        subs.append(sb)
        hs.append(api.batch_upload(sb, params))
        api.batch_set_timing(hs[-1], True)
    pending = [False] * len(hs)
    step_no = [0]

    def run_steps(n):
        for _ in range(n):
            b0 = (step_no[0] % B) * S
            step_no[0] += 1
            for i in range(b0, b0 + S):
                if pending[i]:
                    api.batch_sync(hs[i])
                api.batch_run(hs[i])
                pending[i] = True
            if lockstep:
                for i in range(b0, b0 + S):
                    api.batch_sync(hs[i])
                    pending[i] = False
        for i, h in enumerate(hs):
            if pending[i]:
                api.batch_sync(h)
                pending[i] = False
    try:
        # (a launch plan settles over the shape's first runs: the library leaves a launch out only behind CLEAN_MIN = 3 runs in a
        # row that did not need it)
        run_steps(max(warmup, B, 4))
        for h in hs:
            api.batch_kernel_ms_mean(h, reset=True)
        t0 = time.perf_counter()
        run_steps(steps)
        dt = time.perf_counter() - t0
        km = np.array([api.batch_kernel_ms_mean(h)[0] for h in hs]).mean(axis=0)
        parts = [api.batch_fetch(h) for h in hs]
        profs = [api.batch_profile(h) for h in hs]               # (the report's counters: no profiling switch needed for these)
    finally:
        for h in hs:
            api.batch_free(h)
    res = concat_results(parts)
    by_kernel = {k: 0.0 for k in KERNELS}
    for sb, pr in zip(subs, parts):
        for k, v in pr.algorithmic_bytes_by_kernel(sb, K).items():
            by_kernel[k] += v / len(parts)
    dom = int(np.argmax(km[:3]))
    achieved = by_kernel[KERNELS[dom]] / (km[dom] * 1e-3) / 1e9
    whole = (sum(sb.algorithmic_input_bytes() for sb in subs) + res.algorithmic_output_bytes(K)) / B / (dt / steps) / 1e9
    out = {"workload": "%s: %d regions x %s reads x %d bp, K=%d%s" % (name, regions, "%d-%d" % cfg["n_reads"] if cfg["n_reads"][0] != cfg["n_reads"][1]
                                                                         else str(cfg["n_reads"][0]), cfg["read_len"], K,
                                                                         ", %g of the events tandem duplications (alignment fallback, indelope.nim:312-372)" % dup_frac if dup_frac else ""),
           "value": round(regions * steps / dt, 1), "unit": "regions/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
           "submission": "%d resident batch%s take turns%s%s" % (B, "es" if B > 1 else "", ", %d sub-batches per step" % S if S > 1 else "", ", every step waited for" if lockstep else ""),
           "kernel_ms": dict({k: round(float(v), 4) for k, v in zip(KERNELS, km[:3])}, k_fallback=round(float(km[3]), 4)),
           "roofline": {"kernel": KERNELS[dom], "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "whole_path_frac": round(whole / HBM_PEAK_GBS, 5)},
           "failed_regions": int((res.status != 0).sum()),
           # regions the packed path handed back to the byte-based passes in the last run (0 = every region of the workload stayed
           # on the packed path), runs repeated because a launch plan was wrong
           "regions_to_byte_passes": int(sum(int(p[23]) for p in profs)), "n_reruns": int(sum(int(p[31]) for p in profs)),
           "events": int(res.n_events), "fallback_events": int((res.events["aligned"] == 1).sum()) if res.n_events else 0}
    if check:
        import oracle
        o = oracle.get()
        usable, _ = _host_cpus()
        t1 = time.perf_counter()
        bad, n = None, 0
        for sb, pr in zip(subs, parts):
            lim = min(sb.n_regions, max(10_000, check_regions // max(1, len(subs))), max(0, check_regions - n))
            if lim == 0:
                break
            exp = o.run_regions_mt(sb.slice(0, lim), o.params(K=K), usable)
            got = pr if lim == sb.n_regions else api.run_regions(sb.slice(0, lim), params)
            bad = bad or BatchResult.first_difference(got, exp)
            n += lim
        out["oracle_check"] = {"regions": n, "identical": bad is None, "first_difference": bad, "seconds": round(time.perf_counter() - t1, 2)}
        assert bad is None, "%s: device results differ from the oracle: %s" % (name, bad)
    return out


# ----------------------------------------------------------------------------------- BASELINE configs[0] (C1)
def c1_leg(api, K=27, reps=7):
    """BASELINE configs[0]: ONE synthetic region, 48 x 150 bp reads, end to end.  `cpu_us_per_region`: the CPU path (C oracle with
    the compiled reference ksw2 where built; the Nim binary cannot be built here), median of `reps`; `gpu_us_per_region`: the same
    region through ihp_run_regions (host arrays in, host results out, one region: pure latency), median of `reps`."""
    import oracle
    from indelope_amd import synth
    from indelope_amd.host import BatchResult
    o = oracle.get()
    b, _ = synth.config("C1")
    used_ref = o.use_reference_ksw(True)
    try:
        exp = o.run_regions(b, o.params(K=K))
        cpu = []
        for _ in range(reps):
            t0 = time.perf_counter()
            o.run_regions(b, o.params(K=K))
            cpu.append(time.perf_counter() - t0)
    finally:
        o.use_reference_ksw(False)
    got = api.run_regions(b, api.params(K=K))
    gpu = []
    for _ in range(reps):
        t0 = time.perf_counter()
        api.run_regions(b, api.params(K=K))
        gpu.append(time.perf_counter() - t0)
    d = BatchResult.first_difference(got, exp)
    return {"workload": "C1: 1 region x 48 reads x 150 bp, end to end (BASELINE configs[0])",
            "cpu_us_per_region": round(float(np.median(cpu)) * 1e6, 1), "cpu_min_us": round(min(cpu) * 1e6, 1),
            "cpu_kind": "C oracle (oracle/), ksw2 = %s; one thread" % ("reference ksw2_extz2_sse.c compiled (oracle/_ref)" if used_ref else "scalar restatement"),
            "gpu_us_per_region": round(float(np.median(gpu)) * 1e6, 1), "gpu_min_us": round(min(gpu) * 1e6, 1),
            "gpu_kind": "ihp_run_regions: upload + run + fetch of the one region (launch latency, not throughput)",
            "reps": reps, "identical": d is None, "first_difference": d}


# ------------------------------------------------------------------------------------------ break-even batch size
def break_even(api, K=27, sizes=(1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024), reps=5):
    """Below which batch size a caller should stay on the CPU seam: n C2-shaped regions (64 x 150 bp) through
    ihp_run_regions -- pageable host arrays in, host results out, everything a first-time caller pays -- against the CPU path on
    ONE thread (what the reference is: `indelope.nim` is single-threaded) and on all the threads this process owns (regions are
    independent, so a caller could also do that).  Median of `reps` per size."""
    import oracle
    from indelope_amd import synth
    o = oracle.get()
    usable, _ = _host_cpus()
    used_ref = o.use_reference_ksw(True)
    rows = []
    try:
        full, _ = synth.config("C2", n_regions=max(sizes))
        for n in sizes:
            b = full.slice(0, n)
            for _ in range(4):                                   # (pools and the launch plan of this shape: a launch is left out only behind
                api.run_regions(b, api.params(K=K))              # three runs in a row that did not need it)
            g, c1, cm = [], [], []
            for _ in range(reps):
                t0 = time.perf_counter(); api.run_regions(b, api.params(K=K)); g.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); o.run_regions(b, o.params(K=K)); c1.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); o.run_regions_mt(b, o.params(K=K), min(usable, n)); cm.append(time.perf_counter() - t0)
            rows.append({"regions": n, "gpu_us": round(float(np.median(g)) * 1e6, 1), "cpu_1thread_us": round(float(np.median(c1)) * 1e6, 1),
                         "cpu_%dthreads_us" % usable: round(float(np.median(cm)) * 1e6, 1)})
    finally:
        o.use_reference_ksw(False)
    key = "cpu_%dthreads_us" % usable
    be1 = next((r["regions"] for r in rows if r["gpu_us"] < r["cpu_1thread_us"]), None)
    bem = next((r["regions"] for r in rows if r["gpu_us"] < r[key]), None)
    return {"workload": "n C2-shaped regions (64 x 150 bp) through ihp_run_regions (pageable arrays in, results out) vs the C oracle with the compiled reference ksw2",
            "rows": rows, "gpu_faster_than_one_cpu_thread_from": be1, "gpu_faster_than_%d_cpu_threads_from" % usable: bem, "cpu_threads": usable}


# ------------------------------------------------------------------------------------------ mixed stream
def mixed_stream(api, n_batches=48, in_flight=2):
    """What a sweep hands over is whatever gen_roi yields (indelope.nim:515-545, 601-603): batches of different shapes, one after
    the other.  C2-, C3-, C5- and dup-shaped batches (different seeds) are pushed through ONE process in rotation, `in_flight`
    at a time; reported: regions/s, runs repeated because a launch plan was wrong (`n_reruns`), the same stream with the plans
    switched off (no_hint=1, no_spec=1), and the rate the single-shape timings predict for this sequence (time-weighted)."""
    from indelope_amd import synth
    shapes = [("C2", dict(synth.CONFIGS["C2"], n_regions=5000), 0.0), ("C3", dict(synth.CONFIGS["C3"], n_regions=5000), 0.0),
              ("C5", dict(synth.CONFIGS["C5"], n_regions=2500), 0.0), ("dup10", dict(synth.CONFIGS["C2"], n_regions=5000, config_id=12), 0.1)]
    hs, meta = [], []
    try:
        for name, cfg, dup in shapes:
            pair = []
            for k in range(2):                                   # two batches of every shape (different regions): the single-shape timing keeps two in flight
                b, _ = synth.generate(first_region=k * cfg["n_regions"], dup_frac=dup, **cfg)
                pair.append(api.batch_upload(b.with_trim_bounds(), api.params(K=cfg["K"])))
            hs.append(pair)
            meta.append((name, cfg["n_regions"]))

        def stream(seq, which=lambda i: 0):
            q = []
            for i, s in enumerate(seq):
                h = hs[s][which(i)]
                api.batch_run(h)
                q.append(h)
                if len(q) == in_flight:
                    api.batch_sync(q.pop(0))
            for h in q:
                api.batch_sync(h)

        def reruns():
            return int(sum(api.batch_profile(h)[31] for pair in hs for h in pair))
        seq = [i % len(shapes) for i in range(n_batches)]
        regions = sum(meta[s][1] for s in seq)
        # single-shape rates, two batches of the shape in flight
        single = {}
        for s, (name, n) in enumerate(meta):
            stream([s] * 4, which=lambda i: i & 1)
            t0 = time.perf_counter()
            stream([s] * 8, which=lambda i: i & 1)
            single[name] = (time.perf_counter() - t0) / 8
        predicted = sum(single[meta[s][0]] for s in seq)
        out = {}
        for label, knobs in (("plans_per_shape", {}), ("no_plans", dict(no_hint=1, no_spec=1))):
            api.debug_set(**knobs)
            stream(seq[:len(shapes) * 2])                        # every shape seen twice
            r0 = reruns()
            t0 = time.perf_counter()
            stream(seq)
            dt = time.perf_counter() - t0
            out[label] = {"regions_per_s": round(regions / dt, 1), "ms_per_batch": round(dt / len(seq) * 1e3, 3), "n_reruns": reruns() - r0}
            api.debug_set()
        out.update({"batches": len(seq), "in_flight": in_flight, "shapes": ["%s x %d regions" % m for m in meta],
                    "single_shape_ms_per_batch": {k: round(v * 1e3, 3) for k, v in single.items()},
                    "predicted_regions_per_s": round(regions / predicted, 1),
                    "mixed_over_predicted": round(predicted / (regions / out["plans_per_shape"]["regions_per_s"]), 3),
                    "rerun_share": round(out["plans_per_shape"]["n_reruns"] / len(seq), 4),
                    "note": "one process, one device; batches resident (inputs uploaded once), run + sync per batch; never `value`"})
        return out
    finally:
        for pair in hs:
            for h in pair:
                api.batch_free(h)


# ------------------------------------------------------------------------- the attached strong-scaling leg
def strong_leg(api, torch, dist, idist, rank, world, regions_total, chunk, steps=2, warmup=1, comm=None):
    """north_star's strong-scaling number in the same invocation as the weak one (`--gpus N`, no --config): the C4 generator's
    first `regions_total` regions split over the ranks (dist.shard_bounds), every rank's share resident in chunks, a step = one pass
    over all of them + THE gather of the per-region records to rank 0.  Every rank calls this; rank 0 gets the record."""
    from indelope_amd import synth
    cfg = dict(synth.CONFIGS["C4"])
    params = api.params(K=cfg["K"])
    bounds = idist.shard_bounds(np.ones(regions_total), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    cuts = list(range(lo, hi, max(1, chunk))) + [hi]
    hs = []
    try:
        for i in range(len(cuts) - 1):
            g = dict(cfg, n_regions=cuts[i + 1] - cuts[i])
            sb, _ = synth.generate(first_region=cuts[i], **g)
            hs.append(api.batch_upload(sb.with_trim_bounds(), params))
        views = []
        for h in hs:
            sptr, sn = api.batch_summary_ptr(h)                  # (the address only: no wait)
            views.append(torch.as_tensor(_DevArray(sptr, sn * 8), device="cuda"))
        summary = views[0] if len(views) == 1 else torch.cat(views)
        sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
        last = [None]

        def step():
            for i, h in enumerate(hs):
                api.batch_run(h)
                if i >= 1:
                    api.batch_sync(hs[i - 1])
            api.batch_sync(hs[-1])
            if len(views) > 1:
                torch.cat(views, out=summary)
                torch.cuda.current_stream().synchronize()
            # THE collective: ihp_dist_gather_records (RCCL behind the C ABI), the records land on rank 0's host in region order
            last[0] = comm.gather_records(summary.data_ptr(), hi - lo, root=0, counts=sizes)
        for _ in range(warmup):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ok = None
        if rank == 0:
            got = last[0][0]
            ok = bool(len(got) == regions_total and (got["status"] == 0).all())          # every region's status
        return {"config": "C4 (BASELINE configs[3] generator), first %d of its 5 000 000 regions" % regions_total, "scaling": "strong",
                "value": round(regions_total * steps / dt, 1), "unit": "regions/s", "ms_per_step": round(dt / steps * 1e3, 3),
                "regions_total": regions_total, "regions_per_gpu": sizes, "steps": steps, "warmup": warmup,
                "chunks_per_rank": len(hs), "all_regions_ok": ok,
                "sharding": "contiguous region ranges (dist.shard_bounds), one gather of the per-region records per step: ihp_dist_gather_records (librccl behind the C ABI)"}
    finally:
        for h in hs:
            api.batch_free(h)


def digest(out):
    """The secondary legs in a few hundred characters, as the LAST key of the line: a record that keeps only the tail of the
    line still carries them.  M = million regions/s; `ok` = bit-identical to the oracle."""
    def m(v):
        return None if v is None else round(v / 1e6, 3)
    d = {"C2_M": m(out["value"]), "ms_per_step": out["ms_per_step"], "blocks_ms": out["blocks"]["ms_per_step"],
         "kernel_ms": [out["kernel_ms"][k] for k in KERNELS], "frac": out["roofline"]["frac"], "traffic_stale": out["roofline"]["traffic_stale"],
         "ok": (out.get("oracle_check") or {}).get("identical"), "ok_regions": (out.get("oracle_check") or {}).get("regions")}
    oc = out.get("other_configs") or {}
    for k, v in oc.items():
        d[k] = [m(v["value"]), v.get("oracle_check", {}).get("identical"), v.get("oracle_check", {}).get("regions")]
    ms = out.get("mixed_stream")
    if ms:
        d["mixed"] = [m(ms["plans_per_shape"]["regions_per_s"]), ms["plans_per_shape"]["n_reruns"], ms["mixed_over_predicted"]]
    e = out.get("e2e")
    if e:
        d["e2e"] = {"one_M": m(e["one_batch_regions_per_s"]), "one2h_M": m((e.get("one_batch_as_two_halves") or {}).get("regions_per_s")), "sus_M": m(e["sustained"]["regions_per_s"]), "sus1t_M": m(e["sustained_one_thread"]["regions_per_s"]),
                    "full_one_M": m(e["full_results"]["one_batch_regions_per_s"]), "full_sus_M": m(e["full_results"]["sustained"]["regions_per_s"]),
                    "full_sus1t_M": m(e["full_results"]["sustained_one_thread"]["regions_per_s"]), "slab_MB": e.get("slab_MB")}
    c = out.get("c1")
    if c:
        d["c1_us"] = [c["cpu_us_per_region"], c["gpu_us_per_region"], c["identical"]]
    fcv = out.get("fallback_curve")
    if fcv:
        d["fb_curve_M"] = {k: m(v[0]) for k, v in fcv["by_dup_fraction"].items()}
    be = out.get("break_even")
    if be:
        d["break_even"] = [be["gpu_faster_than_one_cpu_thread_from"], be.get("gpu_faster_than_%d_cpu_threads_from" % be["cpu_threads"])]
    dp = oc.get("deep")
    if dp:
        d["deep_byte_pass_regions"] = dp.get("regions_to_byte_passes")
    cb = out.get("cpu_baseline")
    if cb:
        d["cpu"] = [cb["value"], cb["cores"], out.get("gpu_over_cpu")]
    if out.get("strong"):
        d["strong_M"] = m(out["strong"]["value"])
    d["sha"] = out["build"]["src_sha16"]
    return d


# ----------------------------------------------------------------------------------------------- launching
def spawn_ranks(n, argv):
    """`--gpus N` without a launcher: start N fresh ranks (one per GPU) from a parent that has not touched the GPU and
    relay rank 0's JSON line.  torch.distributed.run gives every child RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["IHP_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in pr.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line:
        print(line)
    return pr.returncode if line or pr.returncode else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, help="BASELINE config id (C2, C3, C4, C5); default C2 -- and, with more than one GPU, "
                                                     "a strong-scaling C4 leg attached as `strong`")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank holds --regions regions of the config (default: the config's own count, at most "
                         "200 000); strong: --regions is the TOTAL (default: the config's own count, e.g. 5 000 000 for C4), split "
                         "over the ranks by dist.shard_bounds")
    ap.add_argument("--regions", type=int, default=0, help="regions per GPU (weak) or in total (strong)")
    ap.add_argument("--chunk", type=int, default=156_250,
                    help="strong scaling: a rank walks its shard in resident chunks of this many regions (inputs of every chunk "
                         "stay in HBM; when the shard's results would not fit beside them only one chunk's results are kept)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-check", action="store_true", help="skip the full-batch oracle comparison after the timed loop")
    ap.add_argument("--quals", action="store_true",
                    help="hand the base qualities to the device and trim there (indelope.nim:23-38) instead of the "
                         "stager-side trim bounds that SURVEY.md 8b/8d specify as the batch input")
    ap.add_argument("--sub-batches", type=int, default=None,
                    help="(default 1; 2 with --in-flight 1) weak scaling: the step submits the batch as this many sub-batches of consecutive regions, each an "
                         "ihp_batch on its own stream: while one sub-batch's last regions drain a kernel, the other's next kernel "
                         "fills the chip.  With more than one chain the per-launch kernel_ms come from device wall-clock stamps "
                         "(ihp_batch_kernel_ms), not from event intervals.  1 = one launch chain, timed with HIP events")
    ap.add_argument("--payload", action="store_true",
                    help="with N > 1 ranks: every step also packs the results on the device and sends each rank's slab to rank 0 "
                         "(the variable-length half of the SURVEY 8e gather)")
    ap.add_argument("--dup-frac", type=float, default=0.0,
                    help="fraction of planted events that are tandem duplications (these send the k-mer tally to the "
                         "alignment fallback, indelope.nim:312-372); 0 = the BASELINE workload")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks form a gloo group, do the per-step gather on host tensors and rank "
                         "0 prints a line with n_gpus = the number of ranks (tests/test_bench_launch.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="form the RCCL process group, run the per-step gather (and --payload) even with ONE rank: the multi-GPU code "
                         "path on a single GPU (tests/test_gpu_round3.py)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="rank 0 checks the gathered per-region records (and payload slabs) against its own fetched results")
    ap.add_argument("--no-other", action="store_true", help="skip the short C3 / C5 / dup10 legs (`other_configs`), `mixed_stream` and `c1` of the default run")
    ap.add_argument("--c3-regions", type=int, default=200_000, help="regions of the C3 leg in `other_configs` (BASELINE configs[2]: 200 000)")
    ap.add_argument("--strong-regions", type=int, default=1_250_000,
                    help="regions of the attached strong-scaling leg (C4 generator; 5 000 000 = all of BASELINE configs[3])")
    ap.add_argument("--in-flight", type=int, default=None, help="weak mode: resident batches of the configured size the stream alternates between -- a step is one pass "
                    "over ONE of them, so this many steps are in flight (1 = rounds 1-3: one batch, split by --sub-batches).  Default: 2 when a "
                    "batch is one launch chain's worth of work (up to ~120 MB of read bases: C2), otherwise 1 with two sub-batches -- two chains "
                    "in flight either way (measured both ways for C3 and C5: DESIGN.md 5)")
    ap.add_argument("--settle-s", type=float, default=0.1, help="weak mode: seconds of untimed steps in front of the W warm-up steps (clocks, launch plans)")
    ap.add_argument("--gather-per-step", action="store_true",
                    help="weak mode with N > 1 ranks: one RCCL gather of the step's R per-region records after EVERY step (round 3's form, "
                         "kept so that rounds can be compared); default: the records of every region a timed block processed "
                         "(steps x R per rank) are kept on the device and gathered ONCE when the block ends")
    ap.add_argument("--lockstep", action="store_true", help="weak mode: wait for every sub-batch of a step before the next step starts (rounds 1-3); default: a sub-batch "
                    "is run again as soon as its own last run is done")
    ap.add_argument("--profile", action="store_true", help="per-phase cycle counters of the kernels (ihp_debug_set profile) in `profile_cycles`")
    ap.add_argument("--knob", action="append", default=[], metavar="KEY=VALUE",
                    help="library path / occupancy switches (ihp_debug_set), e.g. --knob asm_v1=1; results do not depend on them")
    args = ap.parse_args()
    config_given = args.config is not None
    if not config_given:
        args.config = "C2"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent starts the ranks and exits with their code: nothing below (torch, HIP) runs in it
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if args.dry_run:
        import torch.distributed as dist
        from indelope_amd import dist as idist
        if world > 1:
            dist.init_process_group("gloo")
        strong = args.scaling == "strong"
        total = args.regions or 1000
        bounds = idist.shard_bounds(np.ones(total), world) if strong else np.arange(world + 1) * total
        n = int(bounds[rank + 1] - bounds[rank])
        local = torch.full((n, idist.SUMMARY_WORDS), rank, dtype=torch.int32)
        got = idist.gather_summaries(local, rank, world, dst=0)
        strong_rec = None
        if world > 1 and not config_given and not strong:
            # the attached strong-scaling leg (C4 generator, --strong-regions in total): its shard bounds and its one gather
            sb_ = idist.shard_bounds(np.ones(args.strong_regions), world)
            n2 = int(sb_[rank + 1] - sb_[rank])
            got2 = idist.gather_summaries(torch.full((n2, idist.SUMMARY_WORDS), rank, dtype=torch.int32), rank, world, dst=0)
            if rank == 0:
                assert got2.shape[0] == args.strong_regions
                strong_rec = {"config": "C4", "scaling": "strong", "regions_total": int(sb_[-1]), "dry_run": True}
        if rank == 0:
            assert got.shape[0] == int(bounds[-1]) and [int(got[int(bounds[r])][0]) for r in range(world)] == list(range(world))
            print(json.dumps({"metric": "dry run (launcher check, no GPU work)", "value": 0.0, "n_gpus": world, "dry_run": True,
                              "scaling": args.scaling, "regions_total": int(bounds[-1]), "strong": strong_rec, "launched_by": "bench.py --gpus" if
                              os.environ.get("IHP_BENCH_SPAWNED") else "external launcher"}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                    # --force-dist without a launcher: a group of one
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import indelope_amd
    from indelope_amd import synth
    from indelope_amd import dist as idist
    api = indelope_amd.api()
    api.init(local_rank)
    # the data path's one collective goes through the library's own entry points (ihp_dist_*: librccl behind the C ABI, what the
    # Nim host binds); torch.distributed stays for the launch, the barriers around the timed region and the max over ranks
    comm = idist.Communicator(api, rank, world, idist.torch_id_exchange(device="cuda")) if use_dist else None
    if args.profile:
        api.debug_set(profile=1)
    for kv in args.knob:
        k_, v_ = kv.split("=")
        api.debug_set(**{k_: int(v_)})

    cfg = dict(synth.CONFIGS[args.config])
    K = cfg["K"]
    strong = args.scaling == "strong"
    params = api.params(K=K)
    if strong:
        total = args.regions or cfg["n_regions"]
        bounds = idist.shard_bounds(np.ones(total), world)            # uniform reads per region in C2/C4/C5
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        R = hi - lo
        cuts = list(range(lo, hi, max(1, args.chunk))) + [hi]
        S = len(cuts) - 1
    else:
        R = args.regions or min(cfg["n_regions"], 200_000)
        total = R * world
        if args.gather_per_step:
            args.in_flight = 1                               # round 3's form: one resident batch, a gather behind every step
        if args.in_flight is None:
            args.in_flight = 2 if R * 0.5 * (cfg["n_reads"][0] + cfg["n_reads"][1]) * cfg["read_len"] <= 1.2e8 else 1
        B = max(1, args.in_flight)                           # resident batches of R regions; a step is one pass over one of them
        if args.sub_batches is None:
            args.sub_batches = 2 if B == 1 else 1
        lo = rank * B * R
        S = max(1, min(args.sub_batches, R))
        cuts = [lo + b * R + R * i // S for b in range(B) for i in range(S)] + [lo + B * R]
    if strong:
        B = 1
    gen = dict(cfg)
    subs, hs = [], []
    # strong scaling on few GPUs: the results of a whole shard may not fit beside its inputs -> one chunk's results at a time
    # (decided before the uploads: a chunk hands its scratch and result buffers back as soon as it is resident)
    import ctypes
    cu_, ws_, hbm_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    api.b.device_info(ctypes.byref(cu_), ctypes.byref(ws_), ctypes.byref(hbm_))
    hbm = hbm_.value
    stream_outputs = strong and R * 64 * cfg["read_len"] * 9.0 > 0.7 * hbm
    NH = B * S if not strong else S                      # launch chains (ihp_batch handles): handle b * S + i = sub-batch i of batch b
    for i in range(NH):
        gen["n_regions"] = cuts[i + 1] - cuts[i]
        sb, _ = synth.generate(first_region=cuts[i], dup_frac=args.dup_frac, **gen)
        if not args.quals:
            sb = sb.with_trim_bounds()                       # A0 on the host (SURVEY 8a row A0, 8b "trim bounds (a,b)")
        hs.append(api.batch_upload(sb, params))
        if stream_outputs:
            api.batch_release_outputs(hs[-1])
        subs.append(sb if (not strong or (rank == 0 and i == 0)) else None)     # strong: keep one chunk for the host-side legs
    # Stage times come from device wall-clock stamps (with several chains in flight a kernel can wait for wave slots, and an
    # event interval would include that wait); the stamps land in page-locked host memory and ihp_batch_sync adds them up, so
    # the timed loop below reads nothing
    for h in hs:
        api.batch_set_timing(h, True)
    views_all = []
    for h in hs:
        sptr, sn = api.batch_summary_ptr(h)                  # (the address only: no wait)
        views_all.append(torch.as_tensor(_DevArray(sptr, sn * 8), device="cuda") if sn else torch.zeros(0, dtype=torch.int32, device="cuda"))
    views = views_all if strong else views_all[:S]
    summary = views[0] if S == 1 else torch.cat(views)       # per-region records of the rank's regions (strong) / of one step (weak), region order
    # Weak mode gathers what a job of `steps` batches gathers at its end: the records of EVERY region the timed block processed
    # (steps x R per rank).  A run's records are copied into the block's buffer when the run is waited for -- before the
    # batch is run again and overwrites them -- and the buffer goes to rank 0 in ONE gather when the block ends.
    per_block = use_dist and not strong and not args.gather_per_step
    blockbuf = torch.empty(args.steps * R * idist.SUMMARY_WORDS, dtype=torch.int32, device="cuda") if per_block else None
    block_cur, block_slots = [0], {}
    # shard sizes are known to every rank (bounds / equal shares): shards are padded to the longest, ONE gather per step / block
    shard_sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)] if strong else [R * (args.steps if per_block else 1)] * world
    send = blockbuf if per_block else summary
    last_gather, last_payload = [None], [None]

    def wait(i):
        """Wait for handle i's run; in weak mode its records join the block's buffer (they are final now, and the next run of the
        batch overwrites them)."""
        api.batch_sync(hs[i])
        pending[i] = False
        if per_block:
            v = views_all[i]
            if block_cur[0] + v.numel() > blockbuf.numel():
                block_cur[0] = 0                                 # (untimed blocks of another length wrap around)
            blockbuf[block_cur[0]:block_cur[0] + v.numel()].copy_(v)
            # the copy runs on torch's stream and the batch's next run -- on the batch's OWN stream -- rewrites `summary` with its
            # k_summary: the copy has to be over before that run is enqueued (it used to be ordered by timing only)
            torch.cuda.current_stream().synchronize()
            block_slots[i] = block_cur[0]
            block_cur[0] += v.numel()

    def step():
        if strong:
            # chunk after chunk with TWO in flight: a chunk's launch chain ends in tails that the next chunk's first kernels
            # fill; the per-region records stay on the device.  (A chunk's scratch can only go back to the pool once it is done.)
            for i, h in enumerate(hs):
                api.batch_run(h)
                if i >= 1:
                    api.batch_sync(hs[i - 1])
                    if stream_outputs:
                        api.batch_release_outputs(hs[i - 1])
            api.batch_sync(hs[-1])
            if stream_outputs:
                api.batch_release_outputs(hs[-1])
        else:
            # a step is one pass over ONE of the B resident batches (its S sub-batches, each a launch chain on its own stream); the
            # batches take turns.  A batch is run again as soon as ITS last run is done (run -> sync -> run per batch), so B steps
            # are in flight and no step waits for the one before it; block() waits for everything before the clock stops.
            # --lockstep: every step is waited for before the next one starts (rounds 1-3).
            b0 = (step_no[0] % B) * S
            step_no[0] += 1
            for i in range(b0, b0 + S):
                if pending.get(i):
                    wait(i)
                api.batch_run(hs[i])
                pending[i] = True
            if args.lockstep or (use_dist and args.gather_per_step):
                for i in range(b0, b0 + S):
                    wait(i)
            if use_dist and args.gather_per_step:
                gather()                                         # (one resident batch in this form: `summary` is its records)
        if use_dist and strong:
            gather()                                         # strong scaling: the job IS one pass, its gather belongs to the step

    def gather():
        # THE collective of the path: the fixed-size per-region records of every rank to rank 0 (+ the result slabs with --payload)
        if not per_block and S > 1:
            torch.cat(views, out=summary)
            torch.cuda.current_stream().synchronize()
        # (every run whose records are gathered has been waited for: ihp_batch_sync in wait() / step())
        last_gather[0] = comm.gather_records(send.data_ptr(), shard_sizes[rank], root=0, counts=shard_sizes)
        if args.payload and not stream_outputs:
            for h in (hs if strong else hs[:S]):
                last_payload[0] = comm.gather_payload(h, root=0)

    pending = {}
    step_no = [0]

    def block(n):
        # weak scaling: every rank walks through its own batches; the real job gathers ONCE at its end, so a timed
        # block ends with one gather, not one per step (round 3 synchronised all ranks after every 1.8 ms step for no reason)
        block_cur[0] = 0
        for _ in range(n):
            step()
        for i, h in enumerate(hs):                           # (pipelined steps: the last runs of the block)
            if pending.get(i):
                wait(i)
        if use_dist and not strong and not args.gather_per_step:
            gather()

    if not strong:
        # Before the W warm-up steps: every resident batch twice (a batch's launch plan comes from the last finished run of its
        # shape), then ~0.1 s of steps -- a first timed block right behind a few milliseconds of warm-up ran 10 %
        # slower than the two behind it in one run of three (clocks still ramping); untimed, reported as `settle_steps`.
        t_s = time.perf_counter()
        block(2 * B)
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t_s) / (2 * B), 1e-5)
        settle_steps = 2 * B + min(400, int(args.settle_s / est))
        block(settle_steps - 2 * B)
    else:
        settle_steps = 0
    block(args.warmup)
    for h in hs:
        api.batch_kernel_ms_mean(h, reset=True)              # the warm-up runs do not count
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block(args.steps)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # two more blocks of the same K steps, bracketed the same way: `value` is the first block (the contract's K steps), `blocks`
    # says how far a single 40 ms block is from the median
    block_ms = [dt / args.steps * 1e3]
    if not strong:
        for _ in range(2):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            block(args.steps)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            if use_dist:
                t = torch.tensor([d1], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d1 = float(t.item())
            block_ms.append(d1 / args.steps * 1e3)
    # per LAUNCH (a launch processes one sub-batch / chunk): the mean over the timed steps' launches
    km = np.array([api.batch_kernel_ms_mean(h)[0] for h in hs]).mean(axis=0)
    stage = np.array([km[0], km[1], km[2], km[:4].sum()])
    fb_ms = float(km[3])
    strong_rec = None
    if use_dist and world > 1 and not config_given and not strong:
        strong_rec = strong_leg(api, torch, dist, idist, rank, world, args.strong_regions, args.chunk, comm=comm)

    if rank == 0:
        from indelope_amd.host import concat_results
        if stream_outputs:                                   # results were released chunk by chunk: rerun the first chunk for the report
            api.batch_run(hs[0])
            api.batch_sync(hs[0])
        keep = [i for i in range(NH) if subs[i] is not None]
        parts = [api.batch_fetch(hs[i]) for i in keep]
        res = concat_results(parts)
        res_first = res if (strong or B == 1) else concat_results(parts[:S])   # what the gathered records describe
        assert (res.status == 0).all(), "regions failed on the device"
        # B = sum_reads(len+9) + len_refwindow + sum_contigs(5 len+16) + sum_aln(44+4 n_cigar) + sum_events(2K+12)
        n_kept = sum(subs[i].n_regions for i in keep)
        alg_bytes_kept = sum(subs[i].algorithmic_input_bytes() for i in keep) + res.algorithmic_output_bytes(K)
        alg_per_region = alg_bytes_kept / max(n_kept, 1)
        alg_bytes = alg_per_region * R                       # this rank's share per step (exact when every chunk is kept)
        # the same terms split by the kernel that moves them, per LAUNCH: a launch processes one sub-batch / chunk
        by_kernel = {k: 0.0 for k in KERNELS}
        for i, pr in zip(keep, parts):
            for k, v in pr.algorithmic_bytes_by_kernel(subs[i], K).items():
                by_kernel[k] += v / len(keep)
        by_kernel = {k: int(v) for k, v in by_kernel.items()}
        dom = int(np.argmax(stage[:3]))
        achieved = by_kernel[KERNELS[dom]] / (stage[dom] * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel: NOT measured in this process (PMC counters need rocprofv3 passes of their
        # own, tools/profile_round.sh); the figure of the committed passes over the same workload is quoted with its source
        traffic, traffic_source, issue = None, None, None
        pmc, mix = os.path.join(ROOT, PMC_FILE), os.path.join(ROOT, MIX_FILE)
        same = args.config == "C2" and R == 10_000 and S == 1 and B == 2 and not strong
        # the launches a stage consists of (the assembly stage is the packed read phase, the combine phase and the byte-based
        # passes behind them; the ksw2 stage is its plan, the pair sweep and the single sweep; the empty ones count too)
        members = STAGE_MEMBERS[KERNELS[dom]]
        base = lambda n: n.split("<")[0].split("(")[0].split("::")[-1].strip()
        sha = src_sha16()
        stale = None
        if same and os.path.exists(pmc):
            pj = json.load(open(pmc))
            k = pj["kernels"]
            stale = pj.get("src_sha16") != sha
            t = [v["traffic"] * v.get("launches_per_stage", 1) for n, v in k.items() if base(n) in members and v.get("traffic")]
            if t:
                traffic, traffic_source = int(sum(t)), PMC_FILE + (" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same workload; "
                                                                   "2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950; sum over the stage's launches; "
                                                                   + ("TAKEN FROM ANOTHER BUILD of the library (src_sha16 differs): stale)" if stale else "same sources as this run)"))
        issue_by_stage = None
        if same and os.path.exists(mix):
            try:
                mj = json.load(open(mix))
                issue_by_stage = {}
                for stage_name in KERNELS:
                    va = sa = vb = sb = cyc = 0.0
                    for n, v in mj.items():
                        if not isinstance(v, dict):
                            continue
                        if base(n) in STAGE_MEMBERS[stage_name] and v.get("SQ_INSTS_VALU") and v.get("GRBM_GUI_ACTIVE"):
                            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: CU-cycles of a launch = (GUI / 8) x 256 CUs.  The SQ_ACTIVE_INST_*
                            # / SQ_INST_CYCLES_* counters are in quad-cycles per SIMD, i.e. in CU-cycles once summed over a CU's 4 SIMDs
                            va += float(v["SQ_INSTS_VALU"]); sa += float(v.get("SQ_INSTS_SALU", 0)); cyc += float(v["GRBM_GUI_ACTIVE"]) * 32
                            vb += float(v.get("SQ_ACTIVE_INST_VALU", 0)); sb += float(v.get("SQ_INST_CYCLES_SALU", 0))
                    if cyc:
                        issue_by_stage[stage_name] = {
                            "valu_per_cycle_per_cu": round(va / cyc, 3), "salu_per_cycle_per_cu": round(sa / cyc, 3),
                            "valu_busy": round(vb / cyc, 3), "salu_busy": round(sb / cyc, 3),
                            "issue_frac": round(max(va, sa) / cyc / ISSUE_PEAK, 3), "peak": ISSUE_PEAK,
                            "issue_frac_arch": round(va / cyc / ISSUE_PEAK_ARCH, 3), "peak_arch": ISSUE_PEAK_ARCH}
                issue = dict(issue_by_stage.get(KERNELS[dom], {}), stale=mj.get("_src_sha16") != sha, by_stage=issue_by_stage,
                             source=MIX_FILE + " (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_ACTIVE_INST_VALU, SQ_INST_CYCLES_SALU over GRBM_GUI_ACTIVE/8 x 256 "
                                    "CU-cycles, summed over each stage's kernels); `peak` = 1 VALU + 1 SALU wave-instruction per cycle per CU (what a mixed "
                                    "stream reaches: every VALU instruction then costs the half-rate slot), `issue_frac` the busier of the two units "
                                    "against it; `peak_arch` = 2 VALU per cycle per CU (MI355X_MICROARCH.md, pure full-rate streams only), "
                                    "`issue_frac_arch` the VALU rate against that")
            except Exception:
                pass
        out = {
            "metric": "candidate regions/sec (assemble+ksw2+kmer-genotype), 150bp x 64-read batches",
            "value": round(total * args.steps / dt, 1), "unit": "regions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle_steps,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "blocks": {"ms_per_step": [round(x, 4) for x in block_ms], "median_ms_per_step": round(float(np.median(block_ms)), 4),
                       "min_ms_per_step": round(min(block_ms), 4), "value_median": round(total / (float(np.median(block_ms)) * 1e-3), 1),
                       "value_best": round(total / (min(block_ms) * 1e-3), 1),
                       "note": "`value` is the first block (exactly --steps steps, barrier + synchronize on both sides); the other blocks are bracketed the same way"},
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s x %s reads x %d bp, K=%d, err %g (SURVEY 8d generator, seed 0x1DE10BE^%d)"
                                   % (args.config, ("%d regions in total" % total) if strong else ("%d regions/GPU" % R),
                                      "%d-%d" % cfg["n_reads"] if cfg["n_reads"][0] != cfg["n_reads"][1] else str(cfg["n_reads"][0]),
                                      cfg["read_len"], K, cfg["err_rate"], cfg["config_id"])
                                   + (", %g of events tandem duplications" % args.dup_frac if args.dup_frac else ""),
                       "read_trim": "device, from base qualities" if args.quals else "stager (trim bounds in the batch)",
                       "submission": ("%d resident chunk%s of <= %d regions per rank, one after the other%s"
                                      % (S, "s" if S > 1 else "", args.chunk, "; one chunk's results kept at a time" if stream_outputs else ""))
                       if strong else "%d resident batch%s of %d regions take turns, a step is one pass over one of them%s; %s"
                                      % (B, "es" if B > 1 else "", R, (" as %d sub-batches of consecutive regions, each on its own stream" % S) if S > 1 else "",
                                         "every step is waited for before the next starts (--lockstep)" if args.lockstep else
                                         "a batch is run again as soon as its own last run is done (run, sync, run per batch: %d step%s in flight), every run waited for before the clock stops"
                                         % (B, "s" if B > 1 else "")),
                       "regions_per_gpu": R, "regions_total": total,
                       "sharding": ("contiguous region ranges per rank (dist.shard_bounds), one gather (ihp_dist_gather_records: librccl behind the C ABI, records on rank 0's host) of per-region result "
                                    "records per %s" % ("step" if (strong or args.gather_per_step) else "timed block (the job's one gather at its end: the records of every "
                                                        "region the block processed)") + (" + result slabs to rank 0" if args.payload else "")) if use_dist else "single GPU",
                       "gather": ({"per": "step" if (strong or args.gather_per_step) else "block", "records_per_rank_per_gather": int(send.numel() // idist.SUMMARY_WORDS),
                                   "regions_processed_per_rank_per_gather": int(R if (strong or args.gather_per_step) else R * args.steps),
                                   "bytes_per_rank_per_gather": int(send.numel() * 4)} if use_dist else None)},
            "kernel_ms": dict({k: round(float(v), 4) for k, v in zip(KERNELS + ["total"], stage)}, k_fallback=round(fb_ms, 4)),
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom] + (" (k_prepack + k_asm_reads + k_asm_combine3 + byte-based overflow passes)" if dom == 0 else ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_stale": stale,
                         "algorithmic_bytes_per_launch": int(by_kernel[KERNELS[dom]]),
                         "algorithmic_bytes_per_region": round(by_kernel[KERNELS[dom]] * len(keep) / max(n_kept, 1), 1),
                         "launches_per_step": S,
                         "algorithmic_bytes_by_kernel": by_kernel,
                         "issue": issue,
                         "whole_path": {"algorithmic_bytes_per_step": int(alg_bytes * world),
                                        "achieved": round(alg_bytes * world / (dt / args.steps) / 1e9, 2),
                                        "frac": round(alg_bytes * world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}},
            "strong": strong_rec,
            "build": {"src_sha16": sha},
            "results": {"contigs": int(res.n_contigs), "events": int(res.n_events),
                        "tallied": int((res.events["status"] == 0).sum()),
                        "fallback_events": int((res.events["aligned"] == 1).sum()), "regions_inspected": n_kept},
        }
        if args.verify_gather and use_dist:
            # what rank 0 received in the last step against its own results: the records of the RCCL gather are the per-region
            # summaries of the fetched results, the payload slab unpacks to the same results
            got = last_gather[0][0].view(np.int32).reshape(-1)
            mine = idist.summaries_from_result(res_first).view(np.int32).reshape(-1)
            if per_block:
                # the block's buffer holds one set of records per step; every resident batch's LAST run landed at block_slots[handle]:
                # rank 0's own part of the gather against the fetched results of each of its batches
                ok = len(got) == args.steps * R * idist.SUMMARY_WORDS * world
                for b_ in range(B):
                    part = idist.summaries_from_result(concat_results(parts[b_ * S:(b_ + 1) * S])).view(np.int32).reshape(-1)
                    at = block_slots.get(b_ * S)
                    ok = ok and at is not None and bool(np.array_equal(got[at:at + len(part)], part))
            else:
                # (strong scaling keeps only the first chunk's results on the host: its records are the head of the gathered ones)
                ok = bool(np.array_equal(got[:len(mine)], mine)) and (len(got) == len(mine) if (world == 1 and not strong) else len(got) >= len(mine))
            chk = {"records": int(len(got) // idist.SUMMARY_WORDS), "regions_processed": int((R * args.steps if per_block else R if not strong else total) * (world if not strong else 1)),
                   "records_identical_to_own_results": ok, "backend": "ihp_dist (librccl behind the C ABI); launcher: torch.distributed/" + dist.get_backend()}
            if args.payload and last_payload[0] is not None and not strong:
                from indelope_amd.host import BatchResult
                back, nbytes_ = last_payload[0][0][0], last_payload[0][1]
                d = BatchResult.first_difference(back, parts[S - 1] if not strong else parts[-1])
                chk["payload_bytes"] = int(nbytes_[0])
                chk["payload_identical_to_own_results"] = d is None
                ok = ok and d is None
            out["gather_check"] = chk
            assert ok, "gathered records / payload differ from the rank's own results: %r" % chk
        if args.profile and not stream_outputs:
            out["profile_cycles"] = [int(x) for x in sum(np.array(api.batch_profile(h)) for h in hs)]
        batch0 = subs[keep[0]]
        if not strong and S > 1:                             # the host-side legs work on a whole batch
            g2 = dict(cfg)
            g2["n_regions"] = R
            batch0, _ = synth.generate(first_region=lo, dup_frac=args.dup_frac, **g2)
            if not args.quals:
                batch0 = batch0.with_trim_bounds()
        if world == 1 and not strong:
            # the resident batches of the timed loop have given everything they were kept for (results fetched, counters read): they
            # go before the other legs start -- four streams fewer on the runtime's sixteen hardware queues (`mixed_stream` alone
            # keeps eight batches, sixteen streams, alive)
            for h in hs:
                api.batch_free(h)
            hs = []
        if not args.no_e2e and world == 1:
            # right behind the timed loop (the oracle checks below keep the GPU idle for seconds: clocks and pools would have
            # to come back inside the e2e leg's short timed regions)
            out["e2e"] = e2e_rates(api, batch0, params)
            # ... and what its threads leave behind -- a dozen idle streams, page-locked blocks, pooled device buffers -- is given
            # back before the device-resident legs below: with it in place their two launch chains shared a hardware queue more
            # often than not (C5 2.9 -> 2.6 M regions/s, the lock-step leg 6.7 -> 5.9-6.4 M in the same process; the resident
            # batches of the timed loop keep what they own)
            api.b.shutdown()
            api.init(local_rank)
        if not args.no_check:
            # outside the timed loop: every region the rank kept on the host goes through the oracle (all host threads) and must
            # be bit-identical -- the whole 10 000-region batch at the default workload
            import oracle
            from indelope_amd.host import BatchResult
            o = oracle.get()
            usable, _ = _host_cpus()
            t1 = time.perf_counter()
            nchk, bad = 0, None
            for i, pr in zip(keep, parts):
                lim = min(subs[i].n_regions, max(0, 20_000 - nchk))
                if lim == 0:
                    break
                sub = subs[i] if lim == subs[i].n_regions else subs[i].slice(0, lim)
                exp = o.run_regions_mt(sub, o.params(K=K), usable)
                got = pr if lim == subs[i].n_regions else api.run_regions(sub, params)
                d = BatchResult.first_difference(got, exp)
                nchk += lim
                if d is not None:
                    bad = d
                    break
            out["oracle_check"] = {"regions": nchk, "identical": bad is None, "first_difference": bad,
                                   "seconds": round(time.perf_counter() - t1, 2), "threads": usable,
                                   "what": "contigs, supports, ksw2 records + CIGARs, events, k-mer counts and hit positions against "
                                           "the CPU oracle, after the timed loop"}
            assert bad is None, "device results differ from the oracle: " + str(bad)
        if not args.no_other and args.config == "C2" and not strong and world == 1 and not args.dup_frac and R == 10_000:
            # the other single-GPU configs of BASELINE.json, a few steps each, in the same record (never `value`)
            # (two launch chains in flight either way: two resident batches where a batch is one chain's worth of work -- C2's 10 000
            # regions --, one batch as two sub-batches where half a batch already is -- measured both ways, tools/README.md)
            # C3 at BASELINE configs[2]'s own size, every one of its 200 000 regions through the oracle
            out["other_configs"] = {"C3": quick_config(api, "C3", args.c3_regions, steps=4, warmup=2, check=not args.no_check, in_flight=1, sub_batches=2,
                                                       check_regions=args.c3_regions),
                                    "C5": quick_config(api, "C5", 10_000, steps=8, warmup=4, check=not args.no_check, in_flight=1, sub_batches=2),
                                    "dup10": quick_config(api, "C2", 10_000, steps=6, warmup=4, check=not args.no_check, dup_frac=0.1),
                                    # the headline workload submitted as in rounds 1-3: ONE resident batch, two sub-batches, every step waited for
                                    "C2_one_batch_lockstep": quick_config(api, "C2", 10_000, steps=20, warmup=3, check=False, sub_batches=2, in_flight=1, lockstep=True)}
            # the regions the reference admits above C3's 256 reads (gen_roi: up to 600 per roi, indelope.nim:515): 20 000 `deep` regions,
            # n ~ logU[257, 600]; 6 000 of them through the oracle (a deep region costs the oracle seven C2 regions)
            out["other_configs"]["deep"] = quick_config(api, "deep", 20_000, steps=3, warmup=2, check=not args.no_check, in_flight=1, sub_batches=2, check_regions=6_000)
            # what the headline costs as a function of the share of events that take the alignment fallback (indelope.nim:312-372; the
            # C2 generator plants clean indels: 0 of them do): C2 with 0 / 2 / 5 / 10 / 25 % of its events tandem duplications
            fc = {"0": [out["value"], 0]}
            for f in (0.02, 0.05, 0.25):
                q = quick_config(api, "C2", 10_000, steps=6, warmup=4, check=False, dup_frac=f)
                fc["%g" % f] = [q["value"], q["fallback_events"], q["kernel_ms"]["k_fallback"]]
            d10 = out["other_configs"]["dup10"]
            fc["0.1"] = [d10["value"], d10["fallback_events"], d10["kernel_ms"]["k_fallback"]]
            out["fallback_curve"] = {"workload": "C2 (10 000 regions x 64 x 150 bp, two resident batches) with a share of the planted events made tandem duplications: [regions/s, events through the alignment fallback per batch pair, k_fallback ms per launch]",
                                     "by_dup_fraction": {k: fc[k] for k in sorted(fc, key=float)}}
            out["mixed_stream"] = mixed_stream(api)
            out["c1"] = c1_leg(api)
            out["break_even"] = break_even(api)
        if not args.no_cpu and world == 1:
            full = batch0
            out["cpu_baseline"] = cpu_baseline(full, K)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            ss = out["cpu_baseline"].get("single_socket")
            if ss:
                out["gpu_over_single_socket_cpu"] = round(out["value"] / ss["regions_per_s"], 1)
            # the work-based rate beside the HBM roofline: what the reference algorithm would have executed per second
            w = out["cpu_baseline"]["work_per_region"]
            out["work_rate"] = {"char_compares_per_s": round(w["char_compares"] * out["value"], -6),
                                "ksw2_dp_cells_per_s": round(w["ksw2_dp_cells"] * out["value"], -6),
                                "kmer_steps_per_s": round(w["kmer_steps"] * out["value"], -6)}
        out["digest"] = digest(out)
        print(json.dumps(out))
        sys.stdout.flush()
    for h in hs:
        api.batch_free(h)
    if use_dist:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
