"""bench.py host logic that needs no GPU: `--gpus N` starts N ranks itself and relays rank 0's line; the CPU baseline
uses the CPUs this process owns and scales over them."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_the_ranks():
    out = _run(["--gpus", "2", "--dry-run"])
    assert out["n_gpus"] == 2 and out["launched_by"] == "bench.py --gpus" and out["regions_total"] == 2000
    out = _run(["--gpus", "3", "--dry-run", "--scaling", "strong", "--regions", "1000"])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong" and out["regions_total"] == 1000


def test_more_than_one_gpu_and_no_config_attaches_the_strong_scaling_leg():
    """one `--gpus N` invocation yields the weak C2 line AND north_star's strong-scaling number (C4 generator);
    naming a config keeps it to that config.  The dry run walks through the shard bounds and the gather of both."""
    out = _run(["--gpus", "2", "--dry-run", "--strong-regions", "5000"])
    assert out["scaling"] == "weak" and out["strong"] == {"config": "C4", "scaling": "strong", "regions_total": 5000, "dry_run": True}
    out = _run(["--gpus", "2", "--dry-run", "--config", "C2"])
    assert out["strong"] is None
    out = _run(["--dry-run"])
    assert out["strong"] is None


def test_single_rank_needs_no_launcher():
    out = _run(["--dry-run"])
    assert out["n_gpus"] == 1 and out["launched_by"] == "external launcher"


def test_cpu_baseline_scales_over_the_owned_cpus():
    sys.path.insert(0, ROOT)
    import bench
    from indelope_amd import synth
    usable, info = bench._host_cpus()
    assert 1 <= usable <= (os.cpu_count() or 1) and info["affinity"] >= usable
    b, _ = synth.generate(256, n_reads=(64, 64), err_rate=1e-3, config_id=2)
    cb = bench.cpu_baseline(b, 27, want_seconds=4.0)
    assert cb["cores"] <= usable and cb["curve"][0]["threads"] == 1 and cb["curve"][-1]["threads"] == usable
    # the regions are independent: the curve must scale (256 "cores" gave 9x)
    assert cb["scaling_vs_1thread"] >= 0.5 * min(usable, 8), cb["curve"]
