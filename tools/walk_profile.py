"""where the one-launch walk (k_np_walk) spends its time, from a -DNP_WALK_PROFILE build:
  PSF_NP_PERSIST=1 PSF_LIB=$PWD/tools_amd/lib/libpsf_walkprof.so python3 tools/walk_profile.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools_amd as T
from tools_amd._ffi import lib
psf = T.PSFGPV(T.GadgetParameters.init_default(256, 3329), 1024.0); psf.trap_gen(3, export=False); B, m = 1024, psf.m
dev = torch.device("cuda:0")
u = torch.empty((B, psf.n), dtype=torch.int64, device=dev); e = torch.empty((B, m), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), B, seed=7)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1); torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib().psf_debug_walk_prof(out, 1)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2); torch.cuda.synchronize()
lib().psf_debug_walk_prof(out, 0)
nblk = (m + 63) // 64
nS = B // 4
print(f"per sampler workgroup and block: wait {out[0]/nS/nblk:.0f} ticks, body {out[1]/nS/nblk:.0f}")
print(f"workers: wait {out[2]:.3e} ticks in total, tiles {out[3]:.3e} over {out[4]} tiles = {out[3]/max(out[4],1):.0f} per tile")
print(f"sampler workgroup 0: {out[5]} shader ticks in {out[6]} ticks of the 100 MHz clock = {out[6] / 100.0:.1f} us -> {out[5] / max(out[6], 1) * 100.0:.0f} MHz")
