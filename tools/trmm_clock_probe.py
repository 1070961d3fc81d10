"""Average shader clock while k_trmm_f64 runs at C3, from a -DTRMM_CLOCK_PROBE build (every workgroup adds its clock64 and 100 MHz wall_clock64 ticks):
  cd tools_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DTRMM_CLOCK_PROBE -shared -o ../lib/libpsf_clock_probe.so -x hip psfp.hip psf_host.cpp
  PSF_LIB=$PWD/tools_amd/lib/libpsf_clock_probe.so python tools/trmm_clock_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools_amd as T
from tools_amd._ffi import lib, check

n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
dev = torch.device("cuda:0")
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
e = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), B, seed=3)
out = (C.c_ulonglong * 4)()
psf.enable_timing(True)
for rep in range(4):
    lib().psf_debug_trmm_clk(out, 1)
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=100 + rep); torch.cuda.synchronize()
    lib().psf_debug_trmm_clk(out, 0)
    ms = dict(psf.get_timing())["k_trmm_f64"]
    clk, rt, nwg = out[0], out[1], out[2]
    print(f"rep {rep}: k_trmm_f64 {ms:.3f} ms = {psf.m*(psf.m+1)*B/ms*1e-9:.2f} TFLOP/s; {nwg} workgroups, clock64 ticks / wall_clock64 ticks = {clk/rt:.4f}"
          f" -> {clk/rt*100:.1f} MHz if wall_clock64 runs at 100 MHz; FP64 MFMA peak at that clock = {256*4*32*clk/rt*100e6*1e-12:.2f} TFLOP/s", flush=True)
