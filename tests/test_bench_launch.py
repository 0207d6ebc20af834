"""CPU: bench.py's own launcher for `--gpus N` (no torchrun) and the N > 1 control flow, over gloo with 2 ranks.

`python bench.py --gpus 2 --dry-launch` starts two ranks of bench.py exactly as the GPU run does (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set before anything touches a GPU), the ranks join a gloo group, run the priming / warm-up / timed
loops against a counting stand-in for the engine, reduce elapsed time (MAX) and counters (SUM) and rank 0 prints the one
JSON line the driver parses."""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(__file__), "..")
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_dry_launch_two_ranks_reports_both():
    r = _run(["--gpus", "2", "--dry-launch", "--steps", "5", "--warmup", "2", "--streams", "16"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry"] is True and j["steps"] == 5 and j["warmup"] == 2
    assert j["config"]["frames_per_step"] == 32 and j["streams_locked"] == 32          # 16 streams on each of 2 ranks (SUM)
    assert j["value"] > 0 and j["scaling"] == "weak" and j["roofline"] is None and "cpu_baseline" not in j
    # value = frames all ranks processed / max-over-ranks time: 2 ranks x 16 streams x 5 steps
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 * 5 - 160) < 1e-3 * 160


def test_launcher_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1", "--streams", "4"], env={"DABX_BENCH_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--dry-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_dry_launch_eight_ranks_carries_the_per_rank_evidence():
    """BASELINE configs[4] control flow: 8 ranks (one per GPU of a node), gathered per-rank device ids, rates and elapsed times,
    ranks_joined, the collective backend and a scaling efficiency in rank 0's line -- what lets the driver's SCALE record prove
    that 8 ranks sat on 8 different devices."""
    r = _run(["--gpus", "8", "--dry-launch", "--steps", "4", "--warmup", "1", "--streams", "8"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == j["ranks_joined"] == 8 and j["dry"] is True
    assert [p["rank"] for p in j["per_rank"]] == list(range(8)) and len(j["devices"]) == 8
    assert all(p["frames"] == 8 * 4 and p["elapsed_s"] > 0 and p["frames_per_s"] > 0 for p in j["per_rank"])
    assert j["config"]["frames_per_step"] == 64 and j["streams_locked"] == 64
    assert j["collective_backend"] == "gloo" and 0 < j["scaling_efficiency"] <= 1.0 + 1e-9
    # value = all ranks' frames / the slowest rank's time
    assert abs(j["value"] - 8 * 8 * 4 / max(p["elapsed_s"] for p in j["per_rank"])) <= 0.02 * j["value"]


def test_ranks_sharing_a_device_are_refused():
    r = _run(["--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1", "--streams", "4"], env={"DABX_BENCH_DRY_SAME_DEVICE": "1"}, timeout=120)
    assert r.returncode != 0 and "ranks share a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_times_out_on_a_rank_that_never_joins():
    """A rank that hangs before the rendezvous: the launcher's own timeout ends the run, kills exactly the children it
    started (no process is replaced, nothing is matched by name) and reports it; no JSON line."""
    r = _run(["--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1", "--streams", "4"],
             env={"DABX_BENCH_HANG_RANK": "1", "DABX_BENCH_LAUNCH_TIMEOUT": "12"}, timeout=120)
    assert r.returncode != 0 and "ranks still running after the launch timeout" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
