"""Does RCCL accept two ranks on ONE device on this pool?  (NCCL refuses "duplicate GPU"; if RCCL does too, bench.py --oversubscribe has to stay on gloo.)
   python3 tools/probe_rccl_same_device.py            -- starts two ranks of itself through tools_amd/launch.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "RANK" not in os.environ:
    from tools_amd import launch
    sys.exit(launch.run_ranks([sys.executable, os.path.abspath(__file__)], 2, timeout=120))
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=2)
    t = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over RCCL with both ranks on device 0 -> {t.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as ex:      # noqa
    print(f"rank {rank}: RCCL refused: {type(ex).__name__}: {str(ex)[:400]}", flush=True)
    sys.exit(0)
