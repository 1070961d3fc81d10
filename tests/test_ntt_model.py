"""The wave-level NTT (tools_amd/csrc/psf_ntt_core.hpp) on the CPU: the same templates instantiated over a 64-lane array (tests/ntt_model/), against a
schoolbook product in Z_q[X]/(X^n + 1) -- the exchange schedule of lane and register bits, the zeta indexing, leaf products of degree 1 / 2 / 4, both
Montgomery forms and the bound analysis of the unreduced 16-bit form (every 24-bit multiply asserts its operand ranges)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_ntt_model_equals_schoolbook(tmp_path):
    exe = str(tmp_path / "ntt_model")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "ntt_model", "ntt_model.cpp"),
                           os.path.join(ROOT, "tools_amd", "csrc", "psf_host.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "NTT_MODEL OK" in out.stdout
    assert out.stdout.count(": ok (0 mismatches)") == 32          # 16 shapes x {random, extreme} operands
