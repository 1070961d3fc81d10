"""Cycle breakdown of k_np_sample (workgroup 0, wave 0, summed over the blocks of one samp_p call) from a -DNP_PROFILE build:
  cd tools_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DNP_PROFILE=1 -shared -o ../lib/libpsf_np_profile.so -x hip psfp.hip psf_host.cpp
  PSF_LIB=$PWD/tools_amd/lib/libpsf_np_profile.so python tools/np_profile.py c2        (-DNP_PROFILE=2: prologue / steps / epilogue only, no stamps inside the step loop;
  a stamp costs ~85 ticks, so the per-phase figures of NP_PROFILE=1 are upper bounds)"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools_amd as T
from tools_amd._ffi import lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
if cfg == "c2":
    psf = T.PSFGPV(T.GadgetParameters.init_default(256, 3329), 1024.0); psf.trap_gen(3, export=False); B, m = 1024, psf.m
else:
    n = 256; s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, 3329), s, 1.005); psf.trap_gen(4); B, m = 4096, psf.d
u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
e = torch.empty((B, m), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), B, seed=7)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1); torch.cuda.synchronize()
out = (C.c_longlong * 8)()
ev = (C.c_ulonglong * 4)()
sm = (C.c_ulonglong * 8)(); mx = (C.c_ulonglong * 8)()
lib().psf_debug_np_spread(sm, mx, 1)
sg = (C.c_ulonglong * 8)()
if hasattr(lib(), "psf_debug_np_single"): lib().psf_debug_np_single(sg, 1)
lib().psf_debug_np_prof(out, 1)
lib().psf_debug_np_events(ev, 1)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2); torch.cuda.synchronize()
lib().psf_debug_np_prof(out, 0)
lib().psf_debug_np_events(ev, 0)
lib().psf_debug_np_spread(sm, mx, 0)
if sum(sm):
    nblk = (m + 63) // 64
    G = 1 if B <= 1536 else 2
    nw = ((B + 4 * G - 1) // (4 * G)) * 4 * nblk      # sampler waves x launches
    print("NP_PROFILE=3, every sampler wave: mean per wave and launch | largest single wave, in clock64 ticks")
    for k, nm in [(0, "prologue"), (3, "chain up to the ballots (64 steps)"), (4, "settle"), (7, "generic rounds"), (5, "z + update"), (6, "epilogue")]:
        print(f"  {nm:36s} {sm[k]/nw:10.0f} | {mx[k]:10d}")
    print(f"  {'total':36s} {sum(sm)/nw:10.0f} | slowest wave {out[7]:10d}")
    if hasattr(lib(), "psf_debug_np_single"):
        lib().psf_debug_np_single(sg, 0)
        print(f"  single phases: chain max {sg[4]} ticks, {sg[5]} of {nw*64} above 4096; settle max {sg[0]}, {sg[2]} above 4096; generic max {sg[1]}, {sg[3]} above 4096")
print(f"events over all sampler waves: {ev[0]} wave-steps, {ev[1]} entered the settle loop ({100*ev[1]/max(ev[0],1):.2f} %), {ev[2]} the generic rounds ({100*ev[2]/max(ev[0],1):.2f} %), {ev[3]} had a special centre ({100*ev[3]/max(ev[0],1):.2f} %)")
names_unused = None
names = ["prologue (tables, T rows, block above)", "Philox words of 4 steps", "step: LDS + broadcast + centre", "screen + ballots", "settle", "z, update", "epilogue"]
steps = m
tot = sum(out[:7])
for k, nm in enumerate(names):
    print(f"{nm:42s} {out[k]:12d} cycles  {out[k]/steps:9.1f} per step  {100*out[k]/tot:5.1f} %")
print("total", tot, "clock64 ticks for wave 0 of workgroup 0;", tot / steps, "per step;  slowest wave of any single launch:", out[7], "ticks (wave 0 average per launch:", tot // ((m + 63) // 64), ")")
