"""Batch sharding over ranks: preimages are independent, so a job of `total` rows is cut into
contiguous index ranges, one per rank, and the only exchange is a gather of the result rows to rank 0.
The Philox streams are keyed by the GLOBAL row index, so the gathered matrix equals the single-rank result."""
import ctypes as C

import torch
import torch.distributed as dist


def narrow_rows(src_int64, dst_int32, overflow_flag):
    """int64 -> int32 copy of result rows on the current stream.  Device tensors go through the library's
    psf_narrow_rows_dev (one pass: 8 B read + 4 B written per coordinate, range check fused); host tensors
    (the gloo tests) through torch.  overflow_flag: int32 scalar tensor on the same device, OR-ed with 1."""
    if src_int64.is_cuda:
        from ._ffi import lib, check
        check(lib().psf_narrow_rows_dev(C.c_void_p(src_int64.data_ptr()), C.c_void_p(dst_int32.data_ptr()),
                                        C.c_size_t(src_int64.numel()), C.c_void_p(overflow_flag.data_ptr()),
                                        C.c_int(src_int64.device.index or 0),
                                        C.c_void_p(torch.cuda.current_stream(src_int64.device).cuda_stream)), "narrow_rows")
    else:
        if src_int64.numel():
            overflow_flag |= int(src_int64.abs().amax() >= 2**31)
        dst_int32.copy_(src_int64)


def shard_range(rank, world, per_rank):
    """Weak scaling: every rank owns `per_rank` rows; returns (first_index, count) of global row indices."""
    return rank * per_rank, per_rank


def split_rows(total, rank, world):
    """Strong scaling split of `total` rows into `world` nearly equal contiguous ranges."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_rows(local, dst=0, out=None):
    """Gather equally sized row blocks to `dst` (RCCL on GPUs, gloo on CPU); returns the list on dst, else None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        if out is None:
            out = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, out, dst=dst)
        return out
    dist.gather(local, None, dst=dst)
    return None


class AsyncRowGather:
    """Overlapped gather of result rows to `dst`: step i's rows travel (RCCL over xGMI, or gloo on CPU) while step i+1
    computes.  The rows are narrowed to int32 first (|e_i| <= 6 s r sqrt(m) << 2^31 for every supported parameter set,
    checked on device), which halves the bytes on the links; two staging buffers alternate so that a buffer is only
    rewritten after the gather that reads it has completed."""

    def __init__(self, rows, cols, device, dst=0, depth=2, force=False, alloc_world=0):
        """alloc_world: size the buffers as rank `dst` of a job of that many ranks would (receive blocks beyond the real world size are allocated and never
        written) -- the memory side of an 8-rank job rehearsed on the ranks there are (bench.py --emulate-world)."""
        self.enabled = dist.is_initialized() and (dist.get_world_size() > 1 or force)   # force: exercise the path on one rank
        self.rank = dist.get_rank() if self.enabled else 0
        self.world = dist.get_world_size() if self.enabled else 1
        self.alloc_world = max(self.world, int(alloc_world or 0))
        depth = self.fit_depth(rows, cols, device, depth, self.alloc_world, self.rank == dst) if self.enabled else depth
        self.dst, self.depth = dst, depth
        self.stage = [torch.empty((rows, cols), dtype=torch.int32, device=device) for _ in range(depth)] if self.enabled else []
        self.recv = None
        self.spare = []
        if self.enabled and self.rank == dst:
            self.recv = [[torch.empty((rows, cols), dtype=torch.int32, device=device) for _ in range(self.world)] for _ in range(depth)]
            self.spare = [torch.empty((rows, cols), dtype=torch.int32, device=device) for _ in range(depth * (self.alloc_world - self.world))]
        self.work = [None] * depth
        self.step = 0
        self.overflow = torch.zeros((), dtype=torch.int32, device=device) if self.enabled else None
        # A gloo group with device rows (bench.py --oversubscribe: several ranks rehearsing the N > 1 path on the GPUs there are): the collective runs on pinned host
        # copies of the staging blocks -- gloo has no device gather -- and the received blocks are host tensors.  RCCL groups gather device to device, as before.
        self.host_staged = bool(self.enabled and dist.get_backend() == "gloo" and isinstance(device, torch.device) and device.type == "cuda")
        if self.host_staged:
            self.stage_host = [torch.empty((rows, cols), dtype=torch.int32).pin_memory() for _ in range(depth)]
            if self.rank == dst:
                self.recv = [[torch.empty((rows, cols), dtype=torch.int32) for _ in range(self.world)] for _ in range(depth)]

    @staticmethod
    def bytes_per_depth(rows, cols, world, is_dst):
        """device memory one level of the ring costs on this rank: its own int32 staging block, plus one receive block per rank on dst"""
        return rows * cols * 4 * (1 + (world if is_dst else 0))

    @staticmethod
    def fit_depth(rows, cols, device, depth, world, is_dst, free_bytes=None):
        """Largest ring depth <= `depth` (at least 1) whose buffers fit into 80 % of the free device memory.  At C5 (8192 x 122 980 rows per rank,
        8 ranks) one level is 4.03 GB on a worker and 36.3 GB on rank 0; depth 2 = 72.5 GB beside the 60.6 GB key and ~33 GB of batch buffers."""
        if free_bytes is None:
            free_bytes = torch.cuda.mem_get_info(device)[0] if (isinstance(device, torch.device) and device.type == "cuda") else 1 << 62
        per = AsyncRowGather.bytes_per_depth(rows, cols, world, is_dst)
        return max(1, min(depth, int(0.8 * free_bytes) // max(per, 1)))

    def submit(self, e_int64):
        """Queue the gather of this step's rows; returns immediately."""
        if not self.enabled:
            return
        i = self.step % self.depth
        if self.work[i] is not None:
            self.work[i].wait()                      # the previous user of this staging buffer has been delivered
        narrow_rows(e_int64, self.stage[i], self.overflow)       # int64 -> int32 on the current stream
        if self.host_staged:
            self.stage_host[i].copy_(self.stage[i], non_blocking=True)
            torch.cuda.current_stream(self.stage[i].device).synchronize()
            self.work[i] = dist.gather(self.stage_host[i], self.recv[i] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        else:
            self.work[i] = dist.gather(self.stage[i], self.recv[i] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.step += 1

    def finish(self):
        """Wait for every outstanding gather; returns the last step's gathered list on dst (else None)."""
        if not self.enabled:
            return None
        for w in self.work:
            if w is not None:
                w.wait()
        self.work = [None] * self.depth
        if bool(self.overflow.item()):
            raise OverflowError("a preimage coordinate does not fit int32")
        last = (self.step - 1) % self.depth
        return self.recv[last] if (self.rank == self.dst and self.step > 0) else None
