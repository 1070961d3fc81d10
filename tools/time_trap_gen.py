import time, ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import tools_amd as T
from tools_amd._ffi import lib, check
import torch
torch.cuda.synchronize()
t=time.time(); gp = T.GadgetParameters.init_default(512, 2**30); psf = T.PSFPerturbation(gp, 9.0, 512.0); torch.cuda.synchronize(); print("create", time.time()-t)
for i in range(2):
    t=time.time(); check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "tg"); torch.cuda.synchronize(); print("trap_gen", time.time()-t)
