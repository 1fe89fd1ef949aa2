"""bench.py's multi-rank control flow on CPU (no GPU, gloo): `python bench.py --gpus N` starts N ranks itself, the ranks rendezvous on
127.0.0.1, the timed region is the max over ranks, the units are summed, and a rank count that does not match --gpus is refused.
The workload is bench.py's `selftest_null` (a sleep): this file tests the launcher and the reductions, not a kernel."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, "--workload", "selftest_null", "--steps", "5", "--warmup", "1", "--extra-windows", "1"] + extra,
                          capture_output=True, text=True, env=e, timeout=300)


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_single_rank_line():
    d = _line(_run(["--gpus", "1"]))
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "shard1" and d["steps"] == 5 and d["warmup"] == 1
    assert 1.9 <= d["ms_per_step"] <= 20  # sleep(2 ms) per step
    assert abs(d["value"] - 1000 * 5 / (d["ms_per_step"] * 5e-3) / 1e6) < 1e-6 * d["value"] + 1e-9
    assert len(d["windows"]["ms_per_step"]) == 2 and d["windows"]["ms_per_step"][0] == round(d["ms_per_step"], 4)


def test_gpus_2_launches_two_ranks_itself():
    d = _line(_run(["--gpus", "2"]))
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "shard2" and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 3.9  # rank 1 sleeps 4 ms per step: the timed region is the MAX over ranks
    # units of BOTH ranks over that time
    assert abs(d["value"] - 2 * 1000 * 5 / (d["ms_per_step"] * 5e-3) / 1e6) < 1e-6 * d["value"] + 1e-9


def test_driver_style_launch_matches():
    """the driver's own command line: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE / MASTER_* from the env"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        BENCH, "--gpus", "2", "--workload", "selftest_null", "--steps", "4", "--warmup", "1", "--extra-windows", "0"],
                       capture_output=True, text=True, env=e, timeout=300)
    d = _line(p)
    assert d["n_gpus"] == 2 and d["steps"] == 4


def test_rank_count_mismatch_is_refused():
    p = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
