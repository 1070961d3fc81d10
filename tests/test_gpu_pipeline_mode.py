"""PSF_PIPELINE=1 (two buffer sets, normals + FP64 product of call i+1 overlapped with the sampling stages of call i)
must give the same bits as the default sequential mode; run in a subprocess because the switch is read at handle creation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import tools_amd as T
psf = T.PSFPerturbation(T.GadgetParameters.init_default(8, 64), 3.0, 25.0)
psf.trap_gen(1)
B = 300
dev = torch.device("cuda:0")
u = torch.empty((B, 8), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
psf.uniform_targets_dev(u.data_ptr(), B, seed=3, stream=st)
outs = [torch.empty((B, psf.m), dtype=torch.int64, device=dev) for _ in range(5)]
for i, e in enumerate(outs):                      # five calls in flight, no host synchronisation in between
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=50 + i, stream=st)
assert psf.last_status() == 0
h = hashlib.sha256()
for e in outs:
    h.update(e.cpu().numpy().tobytes())
print(h.hexdigest())
''' % ROOT


def run(mode, **extra):
    from tests.conftest import exp_env
    env = exp_env(PSF_PIPELINE=mode, **extra) if (mode != "0" or extra) else exp_env()      # the plain run is the release library, a switch the experiments build
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()[-1]


def test_pipelined_mode_is_bit_identical():
    assert run("0") == run("1")


def test_the_fp64_product_kernels_give_the_same_bits(monkeypatch, exp_lib):
    """k_trmm_f64_big (one workgroup per CU, accumulators in AccVGPRs: the default), k_trmm_f64_reg (PSF_TRMM_VARIANT=1) and k_trmm_f64 (LDS-staged,
    PSF_TRMM_VARIANT=0) run the same ascending fma chains: x and everything downstream must agree bit for bit, here on keys of several row-blocks
    (m = 932: 8 blocks of 128 with a ragged top; structured: 484 rows = 4 blocks, and with n = 40 an ODD number of row-blocks, so the last 256-row
    tile has an empty lower half) and a batch that leaves part of the last column block empty."""
    import numpy as np
    import tools_amd as T
    for n, structured in ((64, False), (64, True), (40, False)):
        gp = T.GadgetParameters.init_default(n, 128)
        psf = T.PSFPerturbation(gp, 3.0, 300.0, structured=structured)
        psf.trap_gen(4, export=False)
        u = np.random.default_rng(5).integers(0, 128, size=(300, n), dtype=np.int64)
        res = {}
        for v in ("0", "1", "2"):
            monkeypatch.setenv("PSF_TRMM_VARIANT", v)
            res[v] = psf.samp_p_stages(u, seed=9)
        for v in ("0", "1"):
            for key in ("x", "p", "e"):
                assert (res[v][key] == res["2"][key]).all(), (n, structured, v, key)
        assert (psf.f_a(res["2"]["e"]) == u).all()


def test_lock_step_gadget_kernel_is_bit_identical():
    """PSF_GADGET_QUEUE=0 selects the lock-step gadget sampler (one lane per problem, sample_z loop) instead of the
    READY/PENDING queue kernel; same Philox streams, same bits."""
    assert run("0") == run("0", PSF_GADGET_QUEUE="0")


@pytest.mark.gpu
def test_narrow_rows_for_gather():
    """shard.narrow_rows on device tensors: values, odd count, overflow flag."""
    import torch
    from tools_amd.shard import narrow_rows
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for count in (0, 1, 7, 4096 * 105 + 1):
        src = torch.randint(-2**31, 2**31, (count,), dtype=torch.int64, generator=g).to(dev)
        dst = torch.full((count,), 77, dtype=torch.int32, device=dev)
        flag = torch.zeros((), dtype=torch.int32, device=dev)
        narrow_rows(src, dst, flag)
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        assert torch.equal(dst.to(torch.int64), src)
    for bad_pos in (0, 12344, 12345):
        src = torch.zeros((12346,), dtype=torch.int64, device=dev)
        src[bad_pos] = 2**31
        dst = torch.empty((12346,), dtype=torch.int32, device=dev)
        flag = torch.zeros((), dtype=torch.int32, device=dev)
        narrow_rows(src, dst, flag)
        torch.cuda.synchronize()
        assert int(flag.item()) == 1


_RCCL_ONE_RANK = r"""
import os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29641"
import torch, torch.distributed as dist
import tools_amd as T
from tools_amd.shard import AsyncRowGather
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
gp = T.GadgetParameters.init_default(16, 257)
psf = T.PSFPerturbation(gp, 4.0, 40.0, device=0)
psf.trap_gen(seed=2)
B, m, n = 96, psf.m, 16
stream = torch.cuda.current_stream().cuda_stream
u = torch.empty((B, n), dtype=torch.int64, device=dev)
e = torch.empty((B, m), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), B, seed=7, first_index=0, stream=stream)
g = AsyncRowGather(B, m, dev, dst=0, force=True)
keep = []
for i in range(3):
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=100 + i, first_index=0, stream=stream)
    keep.append(e.clone())
    g.submit(e)
out = g.finish()
torch.cuda.synchronize()
assert len(out) == 1 and torch.equal(out[0].to(torch.int64), keep[-1]), "gathered rows differ"
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


@pytest.mark.gpu
def test_async_row_gather_over_rccl_one_rank():
    """The bench's N>1 result path (narrow kernel -> async RCCL gather -> wait) on a one-rank RCCL group."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    try:
        r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("the one-rank RCCL group did not come up within 240 s on this box (environment, not the library)")
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("first", ["lib", "torch"])
def test_library_and_torch_share_one_hip_runtime(first):
    """Either load order must work: the ctypes loader maps torch's bundled HIP runtime before the library so that the process
    never holds two runtimes (tools_amd/_ffi.py, tools/probe_load_order.py)."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_load_order.py"), first], capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("the load-order probe did not finish within 150 s on this box")
    if "Timeout (" in r.stderr:                      # the probe's own watchdog (faulthandler, 90 s): a stalled import, not a wrong result
        pytest.skip("the load-order probe stalled on this box: " + r.stderr[-400:])
    out = r.stdout
    assert r.returncode == 0 and "FAILED" not in out and "lib ok" in out and "torch ok" in out, out + r.stderr[-2000:]
    last = [l for l in out.splitlines() if l.strip().startswith("[")][-1]
    assert last.count("libamdhip64") == 1, out          # exactly one HIP runtime mapped at the end


@pytest.mark.gpu
def test_bench_multi_rank_path_on_a_one_rank_group():
    """bench.py --force-dist walks the N>1 code path (process group, barriers, narrowing + RCCL gather, MAX / MIN reductions) on a
    one-rank group; the JSON line must be the last line of stdout and carry valid = true."""
    import json
    env = dict(os.environ, MASTER_PORT="29577")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--config", "bench64", "--no-cpu-baseline",
                            "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("the one-rank RCCL group did not come up within 240 s on this box (environment, not the library)")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert d["valid"] is True and d["n_gpus"] == 1 and d["steps"] == 2 and "RCCL gather" in d["config"]["parallelism"]
    # --force-dist without WORLD_SIZE goes through bench.py's own launcher (tools_amd/launch.py): the rank is a child process, the line is relayed
    assert d["ranks_seen"] == 1 and "self-spawned" in d["launcher"] and d["ms_per_step_ranks"]["min"] <= d["ms_per_step_ranks"]["max"]


@pytest.mark.gpu
def test_bench_under_an_external_launcher_still_works():
    """python -m torch.distributed.run ... bench.py --gpus 1: WORLD_SIZE is set by the launcher, bench.py must not start ranks of its own."""
    import json
    try:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29581",
                            os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--config", "bench64", "--no-cpu-baseline", "--no-latency", "--steps", "2", "--warmup", "1"],
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        pytest.skip("torch.distributed.run did not come up within 300 s on this box (environment, not the library)")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["valid"] is True and d["ranks_seen"] == 1 and d["launcher"].startswith("external")


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_the_box_has_quickly():
    import time
    from tools_amd import launch
    n = launch.visible_gpu_count()
    assert n >= 1
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)], cwd=ROOT, capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and "not starting any rank" in r.stderr and time.time() - t0 < 60


@pytest.mark.gpu
def test_bench_multi_handle_mode():
    """bench.py --multi-handle: the torch-free N-GPU route (psfp_samp_p_multi, one worker thread per handle) on the GPUs this box has (one)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--multi-handle", "--gpus", "1", "--config", "bench64", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["valid"] is True and d["n_gpus"] == 1 and len(d["handle_windows_ms"]) == 1 and d["handle_windows_ms"][0]["done"] > 0


def test_overlapped_async_calls_return_the_rows_of_synchronous_ones(oracle):
    """psfp_samp_p_async / psfp_wait (host buffers): two calls in flight at once -- rows narrowed to int32 on the device, copied in chunks to pinned memory and
    widened by worker threads while the next call computes -- return exactly the rows of two synchronous calls; a third call waits for the first; sliced
    large calls (>= 2048 rows: all but the last 1024, then the tail) equal the device-pointer path bit for bit."""
    import numpy as np
    import tools_amd as T
    n, q, r, s = 24, 2**10, 4.0, 80.0                      # m = 505: 2600 rows are 1.3 M entries (above the straight-through threshold), 2 slices
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    A, (R, Lp, _) = psf.trap_gen(5)
    Bs = [2600, 2100, 2304]
    us = [oracle.uniform_targets(30 + i, B, n, q) for i, B in enumerate(Bs)]
    sync = [psf.samp_p(u, seed=70 + i, first_index=1000 * i) for i, u in enumerate(us)]
    outs = [np.full((B, psf.m), -7, dtype=np.int64) for B in Bs]
    for i, u in enumerate(us):                              # three calls: the third waits for the first inside the library
        psf.samp_p_async(u, outs[i], seed=70 + i, first_index=1000 * i)
    psf.wait()
    for i in range(3):
        assert (outs[i] == sync[i]).all(), i
    orc = oracle.PSFPerturbation(oracle.gadget_params_default(n, q), r, s)
    orc.load_key(A, R, Lp)
    assert (outs[1][:64] == orc.samp_p(71, us[1][:64], first_index=1000)).all()
    assert (psf.f_a(outs[2]) == us[2]).all() and psf.check_domain(outs[2]).all()
    # a small call between asynchronous ones takes the straight path and waits for what is in flight
    psf.samp_p_async(us[0], outs[0], seed=90)
    small = psf.samp_p(us[1][:3], seed=91)
    assert (small == orc.samp_p(91, us[1][:3])).all()
    psf.wait()
    assert (outs[0] == psf.samp_p(us[0], seed=90)).all()
    psf.close()


@pytest.mark.parametrize("mode", ["sdma", "runtime", "kernel:8"])
def test_every_chunk_transport_of_the_host_path_returns_the_same_rows(oracle, monkeypatch, exp_lib, mode):
    """PSF_HOST_COPY picks how a chunk of narrowed rows crosses PCIe: the DMA engine through the HSA runtime (default, psf_sdma.hpp), the HIP runtime's copies,
    or a copy kernel storing into pinned memory.  Each must hand back the rows of the device-pointer path, with chunks smaller than a call (PSF_HOST_CHUNK_MB=1:
    several chunks per worker, both pinned buffers of a worker in use) and asynchronous calls cut into slices or not."""
    import numpy as np
    import torch
    import tools_amd as T
    monkeypatch.setenv("PSF_HOST_COPY", mode)
    monkeypatch.setenv("PSF_HOST_CHUNK_MB", "1")
    n, q, r, s = 24, 2**10, 4.0, 80.0
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    A, (R, Lp, _) = psf.trap_gen(6)
    B = 2500                                                # 2500 x 505 int32 = 5 MB: five chunks, the last one ragged
    u = oracle.uniform_targets(44, B, n, q)
    dev = torch.device("cuda", 0)
    ud = torch.from_numpy(u.astype(np.int64)).to(dev)
    ed = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
    want = []
    for i in range(3):
        psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=300 + i, first_index=17 * i, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want.append(ed.cpu().numpy().copy())
    for sliced in ("0", "1"):
        monkeypatch.setenv("PSF_HOST_ASYNC_SLICE", sliced)
        outs = [np.full((B, psf.m), -3, dtype=np.int64) for _ in range(3)]
        for i in range(3):
            psf.samp_p_async(u, outs[i], seed=300 + i, first_index=17 * i)
        psf.wait()
        for i in range(3):
            assert (outs[i] == want[i]).all(), (mode, sliced, i)
    assert (psf.samp_p(u, seed=300) == want[0]).all()       # the synchronous form (async + wait, tail slice)
    psf.close()


def test_small_host_calls_through_one_pinned_buffer_equal_the_straight_form(oracle, monkeypatch, exp_lib):
    """A small host-pointer call (u + e <= 1 MiB) stages u, e and the flags through one pinned buffer with kernels in stream order and synchronises once;
    PSF_HOST_STRAIGHT=1 keeps the hipMemcpy form.  Same rows, same status, for the three PSF types, growing and shrinking batches."""
    import numpy as np
    import tools_amd as T
    gp = T.GadgetParameters.init_default(8, 128)
    gp2 = T.GadgetParameters.init_default(24, 2**10)
    mk = [(lambda: T.PSFPerturbation(gp, 3.0, 30.0), 8, 128), (lambda: T.PSFGPV(gp, 30.0 * 3.0), 8, 128), (lambda: T.PSFPerturbation(gp2, 4.0, 80.0), 24, 2**10)]
    for make, nn, qq in mk:
        psf = make()
        psf.trap_gen(9)
        for B in (1, 7, 300, 2):
            u = oracle.uniform_targets(100 + B, B, nn, qq)
            monkeypatch.delenv("PSF_HOST_STRAIGHT", raising=False)
            fast = psf.samp_p(u, seed=40 + B, first_index=3)
            monkeypatch.setenv("PSF_HOST_STRAIGHT", "1")
            straight = psf.samp_p(u, seed=40 + B, first_index=3)
            assert (fast == straight).all(), (type(psf).__name__, B)
            assert (psf.f_a(fast) == u).all()
        monkeypatch.delenv("PSF_HOST_STRAIGHT", raising=False)
        psf.close()


def test_batch_host_calls_of_the_nearest_plane_types_equal_the_straight_form(oracle, monkeypatch, exp_lib):
    """psfgpv_samp_p / psfring_samp_p above 1 MiB: cached device buffers, rows narrowed to int32, one pinned copy, threaded widening -- against the straight form
    (PSF_HOST_STRAIGHT=1: two allocations and two pageable copies per call) and the device-pointer call."""
    import numpy as np
    import torch
    import tools_amd as T
    n, q = 16, 257
    gp = T.GadgetParameters.init_default(n, q)
    psf = T.PSFGPV(gp, 60.0)
    psf.trap_gen(4)
    for B in (1200, 700, 1500):                              # m = 304: 2.9 / 1.7 / 3.6 MB of rows, growing and shrinking
        u = oracle.uniform_targets(9 + B, B, n, q)
        monkeypatch.delenv("PSF_HOST_STRAIGHT", raising=False)
        fast = psf.samp_p(u, seed=5 + B, first_index=11)
        monkeypatch.setenv("PSF_HOST_STRAIGHT", "1")
        straight = psf.samp_p(u, seed=5 + B, first_index=11)
        monkeypatch.delenv("PSF_HOST_STRAIGHT", raising=False)
        assert (fast == straight).all(), B
        assert (psf.f_a(fast) == u).all()
    psf.close()


def test_overlapped_async_calls_of_the_nearest_plane_types(oracle):
    """psfgpv_samp_p_async / psfring_samp_p_async + _wait (gpv.rs:152-161, gpv_ring.rs:160-212 on host buffers): two calls in flight return the rows of two
    synchronous calls, a third waits for the first inside the library; the rows equal the oracle's; a small synchronous call in between waits for what is in flight."""
    import math
    import numpy as np
    import tools_amd as T
    n, q, s = 12, 2**9, 60.0
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (Bt, Gt) = psf.trap_gen(6)
    Bs = [700, 520, 900]
    us = [oracle.uniform_targets(40 + i, B, n, q) for i, B in enumerate(Bs)]
    sync = [psf.samp_p(u, seed=80 + i, first_index=500 * i) for i, u in enumerate(us)]
    outs = [np.full((B, psf.m), -7, dtype=np.int64) for B in Bs]
    for i, u in enumerate(us):
        psf.samp_p_async(u, outs[i], seed=80 + i, first_index=500 * i)
    psf.wait()
    for i in range(3):
        assert (outs[i] == sync[i]).all(), i
    assert (psf.f_a(outs[2]) == us[2]).all() and psf.check_domain(outs[2]).all()
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    orc.load_key(A, Bt, Gt)
    assert (outs[1][:24] == orc.samp_p(81, us[1][:24], first_index=500)).all()
    psf.samp_p_async(us[0], outs[0], seed=95)
    small = psf.samp_p(us[1][:2], seed=96)
    assert (small == orc.samp_p(96, us[1][:2])).all()
    psf.wait()
    assert (outs[0] == psf.samp_p(us[0], seed=95)).all()
    psf.close()
    # the ring type
    nr, qr = 16, 3329
    sr = ((2 * 2 * 1.005 * math.sqrt(nr) + 1) * 2) * 4
    ring = T.PSFGPVRing(T.GadgetParametersRing.init_default(nr, qr), sr, 1.005)
    ring.trap_gen(7)
    ur = [oracle.uniform_targets(50 + i, B, nr, qr) for i, B in enumerate((600, 450))]
    syncr = [ring.samp_p(u, seed=60 + i, first_index=33 * i) for i, u in enumerate(ur)]
    outr = [np.full((u.shape[0], ring.K, ring.n), -7, dtype=np.int64) for u in ur]
    for i, u in enumerate(ur):
        ring.samp_p_async(u, outr[i], seed=60 + i, first_index=33 * i)
    ring.wait()
    for i in range(2):
        assert (outr[i] == syncr[i]).all(), i
    assert (ring.f_a(outr[0]) == ur[0]).all()
    ring.close()


@pytest.mark.gpu
def test_asynchronous_calls_keep_their_own_status_by_ticket(oracle):
    """psfp_async_next_ticket / psfp_wait_ticket (the shim's PendingBatch): psfp_wait reports the FIRST failure of everything outstanding and thereby
    consumes the status of calls the caller did not ask about; a ticket keeps each call's own status.  A key whose factor is scaled up makes |p| >= 2^23 (a sampler
    failure, PSF_ERR_SAMPLER) a matter of the seed: one failing and one passing seed are found with synchronous calls, then issued as two overlapped asynchronous
    calls -- whoever is asked first, each ticket answers for its own call, also after psfp_wait has joined both."""
    import numpy as np
    import tools_amd as T
    n, q, r, s, B = 24, 2**10, 4.0, 80.0, 2200                 # m = 505: 2200 rows take the asynchronous transport also in the synchronous entry point
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    A, (R, Lp, _) = psf.trap_gen(5)
    u = oracle.uniform_targets(3, B, n, q)
    t0 = psf.next_ticket()
    out = np.zeros((B, psf.m), dtype=np.int64)
    psf.samp_p_async(u, out, seed=1)
    assert psf.next_ticket() == t0 + 1
    psf.wait_ticket(t0)                                        # fine, and again after a general wait
    psf.wait()
    psf.wait_ticket(t0)
    with pytest.raises(T.PsfError):
        psf.wait_ticket(t0 + 5)                                # never issued
    good = bad = None
    scale = 2.0**23 / (6.0 * np.abs(Lp).max() * np.sqrt(psf.m))
    for attempt in range(12):                                  # a scale at which failure depends on the seed
        psf.load_key(A, R, Lp * scale)
        res = {}
        for seed in range(8):
            try:
                psf.samp_p(u, seed=seed)
                res[seed] = 0
            except T.PsfError as ex:
                assert ex.status == 9
                res[seed] = 9
        if 0 in res.values() and 9 in res.values():
            good = [k for k, v in res.items() if v == 0][0]
            bad = [k for k, v in res.items() if v == 9][0]
            break
        scale *= 0.7 if all(v == 9 for v in res.values()) else 1.4
    if good is None:
        pytest.skip("no scale with seed-dependent failures found")
    outs = [np.zeros((B, psf.m), dtype=np.int64) for _ in range(2)]
    for order in ((bad, good), (good, bad)):
        ta = psf.next_ticket()
        psf.samp_p_async(u, outs[0], seed=order[0])
        psf.samp_p_async(u, outs[1], seed=order[1])
        want = {ta: 9 if order[0] == bad else 0, ta + 1: 9 if order[1] == bad else 0}
        for t in (ta + 1, ta):                                 # the NEWER call first: joining it joins the older one as well, whose status must survive
            if want[t]:
                with pytest.raises(T.PsfError) as ei:
                    psf.wait_ticket(t)
                assert ei.value.status == 9
            else:
                psf.wait_ticket(t)
        try:
            psf.wait()                                          # nothing outstanding: OK, and the tickets still answer
        except T.PsfError:
            pytest.fail("psfp_wait reported a call that had already been joined")
        for t in (ta, ta + 1):
            if want[t]:
                with pytest.raises(T.PsfError):
                    psf.wait_ticket(t)
            else:
                psf.wait_ticket(t)
    psf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["bench64", "c2"])
def test_two_ranks_rehearse_the_multi_gpu_path_on_the_gpus_there_are(config):
    """`bench.py --gpus 2 --oversubscribe`: bench.py's own launcher starts two ranks that share this box's GPU (RCCL refuses two ranks on one device, so the
    process group is gloo and the gather is staged through pinned host memory): everything of the N > 1 path except the RCCL transport runs with real kernels --
    rank environment, the same key on both ranks, targets and Philox streams by GLOBAL preimage index, barriers, the all-gather of the ranks' clocks, the MIN
    reduction of the validity bit, the overlapped gather.  `--verify-gather`: rank 0 recomputes each rank's last step itself and compares with the gathered rows."""
    import json
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--verify-gather", "--config", config, "--no-cpu-baseline",
                            "--no-latency", "--steps", "3", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=420,
                           env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    except subprocess.TimeoutExpired:
        pytest.fail("the two-rank rehearsal did not finish within 420 s")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["valid"] is True and d["gather_verified"] is True
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["ms_per_step_ranks"]["all"]) == 2
    assert "self-spawned" in d["launcher"] and "oversubscribed" in d and "gloo gather" in d["config"]["parallelism"]
    per = {"bench64": 4096, "c2": 1024}[config]
    assert d["config"]["global_batch"] == 2 * per
