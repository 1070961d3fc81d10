"""Child of tests/test_launcher_cpu.py: one rank of a job started by tools_amd.launch.run_ranks.  Joins a gloo group from the environment the launcher set,
reduces over the ranks and (rank 0) prints one JSON line -- the shape of what bench.py's ranks do, without a GPU."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
if mode == "fail" and rank == world - 1:
    sys.exit(7)                                  # before the rendezvous: the others would wait for this rank for ever
if mode == "hang":
    time.sleep(600)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
every = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(every, t)
print(f"rank {rank} chatter", flush=True)       # non-zero ranks' stdout must not reach the job's stdout
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"ranks_seen": world, "sum": sum(float(x.item()) for x in every), "launched_by": os.environ.get("PSF_LAUNCHED_BY")}), flush=True)
