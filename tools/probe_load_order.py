"""Which HIP runtimes end up in the process, by load order (torch wheels bundle their own libamdhip64 / libhsa-runtime64):
   python tools/probe_load_order.py lib|torch"""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(90, exit=True)      # a hang prints where it is and ends the probe
_t0 = time.time()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
first = sys.argv[1] if len(sys.argv) > 1 else "lib"

def runtimes():
    maps = open("/proc/self/maps").read()
    return sorted({l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "libhsa-runtime" in l})

def use_lib():
    import tools_amd as T
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(8, 64), 3.0, 25.0)
    psf.trap_gen(1)
    return "ok"

def use_torch():
    import torch
    torch.cuda.init()
    return f"ok ({torch.cuda.device_count()} device)"

steps = [("lib", use_lib), ("torch", use_torch)] if first == "lib" else [("torch", use_torch), ("lib", use_lib)]
for name, fn in steps:
    try:
        print(name, fn(), f"[{time.time() - _t0:.1f} s]")
    except Exception as ex:
        print(name, "FAILED:", ex)
    print("  ", runtimes())
