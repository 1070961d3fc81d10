"""One process per GPU without an external launcher: `run_ranks` starts N fresh child processes of a command with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- what `python -m torch.distributed.run` would export -- relays rank 0's
standard output line by line (its JSON line stays the parent's last line), sends the other ranks' output to standard error, and ends the
whole job as soon as one rank fails: the survivors get SIGTERM, then SIGKILL, by the exact PIDs started here.

The parent never touches the GPU: it must not import torch.cuda or load the HIP library before (or after) starting the ranks -- a process
that has initialised HIP may not be replaced (`os.exec*`) on the GPU boxes, and nothing here ever is: children are `subprocess.Popen`.
GPUs are counted from sysfs (`visible_gpu_count`), not through a HIP call.
"""
import glob
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPU agents of this host from the KFD topology (no HIP call), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are
    plain index lists.  0 when the host has no /sys/class/kfd (this container)."""
    count = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as fh:
                for line in fh:
                    key, _, val = line.partition(" ")
                    if key == "simd_count" and int(val) > 0:
                        count += 1
                        break
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            count = min(count, len([x for x in val.split(",") if x.strip() != ""]))
    return count


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "PSF_LAUNCHED_BY": "tools_amd.launch"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def _pump(src, dst, prefix=b""):
    for line in iter(src.readline, b""):
        dst.write(prefix + line)
        dst.flush()
    src.close()


def run_ranks(cmd, world, port=None, timeout=None, grace=5.0, out=None, err=None):
    """Run `cmd` (argv list) as `world` ranks; returns the job's exit code: 0 when every rank returned 0, otherwise the first
    non-zero code seen (a rank killed by signal s counts as 128 + s; a job that exceeds `timeout` seconds as 124)."""
    out = out or getattr(sys.stdout, "buffer", sys.stdout)
    err = err or getattr(sys.stderr, "buffer", sys.stderr)
    port = port or free_port()
    procs, pumps = [], []
    try:
        for r in range(world):
            p = subprocess.Popen(cmd, env=rank_env(r, world, port), stdout=subprocess.PIPE, stderr=None, stdin=subprocess.DEVNULL)
            procs.append(p)
            t = threading.Thread(target=_pump, args=(p.stdout, out if r == 0 else err, b"" if r == 0 else b"[rank %d] " % r), daemon=True)
            t.start()
            pumps.append(t)
        t_end = None if timeout is None else time.monotonic() + timeout
        rc = 0
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = 128 - code if code < 0 else code
                    err.write(b"[launch] rank %d exited with %d: stopping the other ranks\n" % (r, code))
                    err.flush()
            if rc != 0 or (t_end is not None and time.monotonic() > t_end):
                if rc == 0:
                    rc = 124
                    err.write(b"[launch] time limit reached: stopping every rank\n")
                    err.flush()
                break
            time.sleep(0.05)
        return rc
    finally:
        stop(procs, grace)
        for t in pumps:
            t.join(timeout=2.0)


def stop(procs, grace=5.0):
    """SIGTERM to the ranks still alive, SIGKILL after `grace` seconds -- by PID, never by pattern."""
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    t_end = time.monotonic() + grace
    for p in alive:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()
