#!/usr/bin/env python3
"""PSFGPVRing::f_a (gpv_ring.rs:243-247) at C4 (n = 256, q = 3329, 4096 preimages) on device buffers: the k+2 R_q products against the cached images of a
(PSF_RING_FA unset / ntt) or the product with rot^-(iota(a)) on the int8 matrix cores (PSF_RING_FA=matmul).  Run once per setting; prints one JSON line
with the HIP-event time per call and a checksum of u so that the two runs can be compared."""
# the PSF_* switches this script sets are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
import os as _os
_os.environ.setdefault("PSF_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import json
import math
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tools_amd as T

n, q = 256, 3329
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = 20
s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
psf.trap_gen(1)
sg = psf.samp_d(seed=2, B=B)
dev = torch.device("cuda", 0)
d_sg = torch.from_numpy(sg.reshape(B, -1)).to(dev)
d_u = torch.empty((B, n), dtype=torch.int64, device=dev)
d_ok = torch.empty((B,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
psf.f_a_dev(d_sg.data_ptr(), d_u.data_ptr(), d_ok.data_ptr(), B, stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    psf.f_a_dev(d_sg.data_ptr(), d_u.data_ptr(), d_ok.data_ptr(), B, stream=st)
e1.record()
torch.cuda.synchronize()
u = d_u.cpu().numpy()
print(json.dumps({"mode": os.environ.get("PSF_RING_FA", "ntt"), "n": n, "q": q, "B": B, "us_per_call": round(e0.elapsed_time(e1) * 1e3 / reps, 2),
                  "all_in_domain": bool(d_ok.cpu().numpy().all()), "crc32_u": zlib.crc32(u.tobytes())}))
