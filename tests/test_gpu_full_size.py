"""BASELINE.json's full-size configurations through the size-independent properties the domain offers: every preimage of
the batch satisfies A e = u (mod q) and the norm bound of check_domain (mp_perturbation.rs:366-369, :396-402; gpv.rs:190-193,
:219-224; gpv_ring.rs:243-247, :274-283), and a bounded sample of the same batch is bitwise equal to the CPU oracle.
Runs bench.py (the measured path itself) in a subprocess and reads its JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("config,sample", [("c3", 256), ("c2", 64), ("c4", 128)])
def test_full_size_batch_is_valid_and_matches_the_oracle_on_a_sample(config, sample):
    d = run_bench("--config", config, "--steps", "1", "--warmup", "0", "--cpu-sample", str(sample))
    assert d["valid"] is True                                   # A e = u and check_domain for every row of the batch
    assert d["cpu_baseline"]["matches_gpu_bitwise"] is True     # the sampled rows equal the oracle's
    assert d["n_gpus"] == 1 and d["data"] == "synthetic"
    if config == "c3":
        assert d["roofline"]["bound"] == "mfma" and 0.5 < d["roofline"]["frac"] <= 1.0
