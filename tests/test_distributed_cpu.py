"""World-size-2 gloo test of the N>1 path: batch sharding by global preimage index + gather to rank 0
(tools_amd/shard.py, used by bench.py).  The compute engine here is the CPU oracle standing in for the HIP
library (no GPU in this container); what is under test is the index arithmetic and the collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, q, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from tools_amd.shard import shard_range, gather_rows, AsyncRowGather
    n = 5
    psf = O.PSFPerturbation(O.gadget_params_default(n, q), 2.5, 25.0)
    assert psf.trap_gen(8) == 0                         # same key seed on every rank -> same key
    first, count = shard_range(rank, world, per_rank)
    u = O.uniform_targets(1, count, n, q, first_index=first)
    e = psf.samp_p(77, u, first_index=first, nthreads=1)
    got = gather_rows(torch.from_numpy(e), dst=0)
    # the overlapped int32 gather used by bench.py: three "steps" in flight over two staging buffers
    ag = AsyncRowGather(count, psf.m, torch.device("cpu"), dst=0)
    for step in range(3):
        ag.submit(torch.from_numpy(e) + step)
    last = ag.finish()
    if rank == 0:
        np.save(out_path, torch.cat(got).numpy())
        assert (torch.cat(last).to(torch.int64) == torch.cat(got) + 2).all()
    else:
        assert got is None and last is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank(tmp_path, oracle):
    world, per_rank, q = 2, 6, 32
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), per_rank, q, out), nprocs=world, join=True)
    gathered = np.load(out)
    psf = oracle.PSFPerturbation(oracle.gadget_params_default(5, q), 2.5, 25.0)
    assert psf.trap_gen(8) == 0
    u = oracle.uniform_targets(1, world * per_rank, 5, q)
    full = psf.samp_p(77, u, first_index=0, nthreads=1)
    assert gathered.shape == full.shape and (gathered == full).all()


def test_split_rows_covers_everything():
    from tools_amd.shard import split_rows, shard_range
    for total in (0, 1, 7, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            spans = [split_rows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
    assert shard_range(3, 8, 4096) == (3 * 4096, 4096)
