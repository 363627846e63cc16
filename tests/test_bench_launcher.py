"""bench.py's own N-rank launcher, on the CPU: `--gpus 2 --dry --backend gloo` must start two fresh rank
processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), run the rendezvous, the barriers, the max-over-ranks
timing and the per-rank gather, and forward ONE JSON line from rank 0 with n_gpus = 2.  No kernel runs (the
reference has no multi-GPU path to compare with: models/direction_diffusion.py:87 is single-device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_launcher_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["dry"] is True and j["steps"] == 3 and j["warmup"] == 1
    assert j["config"]["rows_per_gpu"] == [15777, 15777]          # one whole SynEgo1000-shaped shard per rank
    assert j["config"]["launched_by"] == "bench.py launcher"
    assert j["value"] is None                                      # a dry run reports no performance number
    assert j["metric"].startswith("latent node-steps/sec") and j["scaling"] == "weak"


def test_single_process_needs_no_launcher():
    r = _run(["--dry", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["rows_per_gpu"] == [15777]
    assert j["config"]["launched_by"] == "single process"


def test_launcher_reports_a_failing_rank():
    """an unknown backend makes every rank fail in init_process_group: the launcher must exit non-zero and print no
    JSON line"""
    r = _run(["--gpus", "2", "--dry", "--backend", "no_such_backend", "--steps", "1"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launcher_refuses_more_gpus_than_visible():
    """without --dry the launcher counts devices (torch.cuda.device_count() initialises nothing) before it starts
    anything; with no or too few GPUs it says so and starts no rank"""
    import torch
    if torch.cuda.device_count() >= 64:
        return
    r = _run(["--gpus", "64", "--steps", "1"])
    assert r.returncode == 2 and "--gpus 64" in r.stderr and r.stdout.strip() == ""


def test_rank_mode_under_an_external_launcher():
    """WORLD_SIZE already set (torch.distributed.run, as the driver starts it): bench.py is a rank, not a launcher"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = []
    for rank in range(2):
        e = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry", "--backend", "gloo", "--steps", "2"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    j = _json_line(outs[0][0])
    assert j["n_gpus"] == 2 and j["config"]["launched_by"] == "external launcher"
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints


def test_a_stopped_secondary_leg_does_not_cost_the_line():
    """rank 1 never joins a collective after the headline (HD_BENCH_DRY_STALL_RANK: the dry run's stand-in for a secondary
    leg that stops with more than one rank): after --legs-timeout seconds every rank's watchdog ends its process, rank 0
    having printed the line with the unfinished leg as an error entry; exit status 0"""
    r = _run(["--gpus", "2", "--dry", "--backend", "gloo", "--steps", "2", "--legs-timeout", "3"],
             env={"HD_BENCH_DRY_STALL_RANK": "1"}, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and "not finished 3 s" in j["dry_stall"]["error"]
    assert "exceeded --legs-timeout" in r.stderr
