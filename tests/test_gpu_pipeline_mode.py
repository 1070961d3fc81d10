"""PSF_PIPELINE=1 (two buffer sets, normals + FP64 product of call i+1 overlapped with the sampling stages of call i)
must give the same bits as the default sequential mode; run in a subprocess because the switch is read at handle creation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import tools_amd as T
psf = T.PSFPerturbation(T.GadgetParameters.init_default(8, 64), 3.0, 25.0)
psf.trap_gen(1)
B = 300
dev = torch.device("cuda:0")
u = torch.empty((B, 8), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
psf.uniform_targets_dev(u.data_ptr(), B, seed=3, stream=st)
outs = [torch.empty((B, psf.m), dtype=torch.int64, device=dev) for _ in range(5)]
for i, e in enumerate(outs):                      # five calls in flight, no host synchronisation in between
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=50 + i, stream=st)
assert psf.last_status() == 0
h = hashlib.sha256()
for e in outs:
    h.update(e.cpu().numpy().tobytes())
print(h.hexdigest())
''' % ROOT


def run(mode):
    env = dict(os.environ, PSF_PIPELINE=mode)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()[-1]


def test_pipelined_mode_is_bit_identical():
    assert run("0") == run("1")
