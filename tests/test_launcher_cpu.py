"""bench.py starts its own ranks (tools_amd/launch.py): `python3 bench.py --gpus N` must be a complete command on an N-GPU box and fail fast,
with a message, on a box with fewer GPUs.  Here, without a GPU: the launcher itself with two gloo children (environment, one JSON line relayed as the last
line of the job's stdout, the other ranks' output kept off it), exit-code propagation with the surviving ranks stopped, the time limit, and bench.py's refusal."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "rank_child.py")
DRIVER = """
import sys
sys.path.insert(0, %r)
from tools_amd import launch
sys.exit(launch.run_ranks([sys.executable, %r] + sys.argv[2:], int(sys.argv[1]), timeout=%s, grace=2.0))
"""


def job(world, *child_args, timeout=None, wall=180):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", DRIVER % (ROOT, CHILD, timeout), str(world), *child_args], capture_output=True, text=True, timeout=wall, cwd=ROOT)
    return r, time.time() - t0


def test_two_ranks_one_json_line_last():
    r, _ = job(2, "ok")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    d = json.loads(lines[-1])
    assert d == {"ranks_seen": 2, "sum": 3.0, "launched_by": "tools_amd.launch"}
    assert not any("rank 1" in line for line in lines)                  # rank 1's stdout went to the job's stderr, prefixed
    assert "[rank 1] rank 1 chatter" in r.stderr


def test_a_failing_rank_fails_the_job_and_stops_the_others():
    r, took = job(2, "fail")
    assert r.returncode == 7, (r.returncode, r.stderr[-1500:])
    assert "rank 1 exited with 7" in r.stderr and took < 60             # rank 0 was waiting at the rendezvous: stopped by the launcher, not by gloo's own time-out


def test_time_limit_stops_every_rank():
    r, took = job(2, "hang", timeout=3.0)
    assert r.returncode == 124 and took < 45, (r.returncode, took, r.stderr[-800:])


def test_bench_refuses_more_gpus_than_the_host_has():
    """(this container has none; on the GPU box the same command with --gpus 9 does the same)"""
    from tools_amd import launch
    n = max(launch.visible_gpu_count(), 1)                              # (--gpus 1 is not a launcher job)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and f"--gpus {n + 1}" in r.stderr and "not starting any rank" in r.stderr, r.stderr[-1500:]
    assert r.stdout.strip() == "" and time.time() - t0 < 60


def test_rank_environment():
    from tools_amd import launch
    env = launch.rank_env(3, 8, 29999, base={"PATH": "/bin"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
