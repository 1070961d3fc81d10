"""The host-pointer entry point (its own non-blocking compute stream) against the device-pointer entry point (the caller's stream) on seeded random PSFPerturbation
configurations, the FIRST call after key generation included: both must return the same rows.  Round 5: the batch buffers were cleared on the null stream without a
barrier behind the clears, so the first kernels of a first host-pointer call could be overwritten by a late clear (one whole-batch mismatch in 240 000 first calls of
tools/host_vs_device_fuzz.py, small keys with 1024 preimages); ensure_batch / ensure_np_batch now drain the device before they return."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("first,wide", [(200000, True), (210000, False)])
def test_host_pointer_calls_equal_device_pointer_calls_from_the_first_call_on(first, wide):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "host_vs_device_fuzz.py"), str(first), "1500"] + (["--wide"] if wide else ["--narrow"]) + ["3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("done:")]
    assert last and " 0 mismatches" in last[-1], r.stdout[-3000:]
