"""Every experiment switch of the PSFPerturbation path (alive in the experiments build only: tools_amd/csrc `make exp`; the release library reads none) selects another kernel, stream arrangement or slicing for the SAME arithmetic: on seeded random shapes
(ragged batches on both sides of the 128 / 256 tile boundaries, key dimensions with partial row blocks, moduli of one to three limbs) each setting must
reproduce the default's bytes.  One subprocess per setting (the switches are read at handle creation / first use); the key comes from the same seed every time,
so key generation under PSF_CHOL's three forms is part of what is compared (A, R bitwise; the factors agree within rounding, checked elsewhere), and the
preimages are compared through the factor of the default form, loaded explicitly."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, hashlib, json, math
sys.path.insert(0, %r)
import numpy as np
import tools_amd as T
case = int(sys.argv[1])
rng = np.random.default_rng(7000 + case)
n = int(rng.integers(6, 48))
q = int(rng.choice([2**int(rng.integers(6, 31)), 12289, 1073741789, 2**45, 2**60]))
k = int(math.ceil(math.log2(q)))
r = float(rng.choice([2.0, 3.0, 4.5]))
m_bar = n * k + int(rng.integers(0, 64))
s = r * math.sqrt(5.0) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0) * 1.5
B = int(rng.choice([1, 127, 129, 255, 256, 257, 300, 512, 700]))
gp = T.GadgetParameters(n, k, m_bar, 2, q)
# the key of the DEFAULT factorisation, produced in this process by a handle that ignores PSF_CHOL ... no: it is passed in on argv[2] (a file) by the driver
key = np.load(sys.argv[2])
psf = T.PSFPerturbation(gp, r, s)
psf.load_key(key["A"], key["R"], key["L"])
u = np.random.default_rng(case).integers(0, min(q, 2**62), size=(B, n), dtype=np.int64) %% q
outs = [psf.samp_p(u, seed=40 + i, first_index=1000 * i) for i in range(3)]
h = hashlib.sha256()
for e in outs:
    h.update(np.ascontiguousarray(e).tobytes())
print(json.dumps({"hash": h.hexdigest(), "shape": [n, q, m_bar, B]}))
''' % ROOT

KEYGEN = r'''
import sys, math
sys.path.insert(0, %r)
import numpy as np
import tools_amd as T
case = int(sys.argv[1])
rng = np.random.default_rng(7000 + case)
n = int(rng.integers(6, 48))
q = int(rng.choice([2**int(rng.integers(6, 31)), 12289, 1073741789, 2**45, 2**60]))
k = int(math.ceil(math.log2(q)))
r = float(rng.choice([2.0, 3.0, 4.5]))
m_bar = n * k + int(rng.integers(0, 64))
s = r * math.sqrt(5.0) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0) * 1.5
psf = T.PSFPerturbation(T.GadgetParameters(n, k, m_bar, 2, q), r, s)
A, (R, L, _) = psf.trap_gen(900 + case)
np.savez(sys.argv[2], A=A, R=R, L=L)
''' % ROOT

SETTINGS = [{}, {"PSF_PIPELINE": "1"}, {"PSF_HALVES": "1"}, {"PSF_HOST_SLICE": "64"}, {"PSF_HOST_SLICE": "200"}, {"PSF_GADGET_QUEUE": "0"}, {"PSF_ROUND": "wave"},
            {"PSF_TRMM_VARIANT": "0"}, {"PSF_TRMM_VARIANT": "1"}, {"PSF_TRMM_GR": "4", "PSF_TRMM_GC": "8"}]


def run(script, case, path, **extra):
    from tests.conftest import exp_env
    env = exp_env(**extra)                       # a switch = the experiments build of the library; no switch = the release library
    r = subprocess.run([sys.executable, "-c", script, str(case), path], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""


@pytest.mark.parametrize("case", range(8))
def test_every_switch_reproduces_the_default_bytes(tmp_path, case):
    path = str(tmp_path / "key.npz")
    run(KEYGEN, case, path)
    base = json.loads(run(SCRIPT, case, path))
    for st in SETTINGS[1:]:
        got = json.loads(run(SCRIPT, case, path, **st))
        assert got["hash"] == base["hash"], (st, base["shape"])


@pytest.mark.parametrize("case", range(2))
def test_key_generation_forms_produce_the_same_key_material(tmp_path, case):
    """A and R are bitwise the same under the three Cholesky forms (they do not depend on the factorisation); the factors agree to rounding."""
    import numpy as np
    keys = {}
    for form in ("gemm", "stream", "right"):
        path = str(tmp_path / f"key_{form}.npz")
        run(KEYGEN, case, path, PSF_CHOL=form)
        keys[form] = np.load(path)
    for form in ("stream", "right"):
        assert (keys[form]["A"] == keys["gemm"]["A"]).all() and (keys[form]["R"] == keys["gemm"]["R"]).all()
        np.testing.assert_allclose(keys[form]["L"], keys["gemm"]["L"], rtol=0, atol=1e-9 * np.abs(keys["gemm"]["L"]).max())
