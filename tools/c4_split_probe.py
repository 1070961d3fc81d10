#!/usr/bin/env python3
"""C4 (PSFGPVRing n = 256, q = 3329, 4096 preimages) as two half batches on two streams: does the FP64-MFMA update phase of one half hide behind the sampling of
the other?  Two handles with the same key (a handle's walk buffers serve one call at a time); rows are the single-handle rows (global preimage index)."""
import math
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tools_amd as T

n, q, B = 256, 3329, 4096
s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
gp = T.GadgetParametersRing.init_default(n, q)
hs = [T.PSFGPVRing(gp, s, 1.005) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
for h in hs:
    h.trap_gen(4)
d = hs[0].d
dev = torch.device("cuda", 0)
u = torch.empty((B, n), dtype=torch.int64, device=dev)
hs[0].uniform_targets_dev(u.data_ptr(), B, seed=7, first_index=0, stream=torch.cuda.current_stream().cuda_stream)
e_one = torch.empty((B, d), dtype=torch.int64, device=dev)
e_two = torch.empty((B, d), dtype=torch.int64, device=dev)
streams = [torch.cuda.Stream() for _ in hs]
torch.cuda.synchronize()


def one(seed):
    hs[0].samp_p_dev(u.data_ptr(), e_one.data_ptr(), B, seed=seed, first_index=0, stream=torch.cuda.current_stream().cuda_stream)


def split(seed, concurrent=True):
    k = len(hs)
    per = B // k
    for i, h in enumerate(hs):
        st = streams[i].cuda_stream if concurrent else torch.cuda.current_stream().cuda_stream
        h.samp_p_dev(u[i * per:].data_ptr(), e_two[i * per:].data_ptr(), per, seed=seed, first_index=i * per, stream=st)


def timeit(f, reps=20):
    f(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        f(100 + r)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t1 = timeit(one)
t2 = timeit(lambda sd: split(sd, True))
t3 = timeit(lambda sd: split(sd, False))
one(5); torch.cuda.synchronize(); split(5, True); torch.cuda.synchronize()
print(f"one call of {B}: {t1:.3f} ms;  {len(hs)} x {B // len(hs)} on {len(hs)} streams: {t2:.3f} ms;  the same back to back on one stream: {t3:.3f} ms;  rows equal: {bool((e_one == e_two).all())}")
print("forms:", [h.nearest_plane_form() for h in hs])
