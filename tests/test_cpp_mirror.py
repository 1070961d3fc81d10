"""The C++ mirror of the PSF trait (include/psf_mi355x.hpp) compiles with plain g++ against the C ABI and runs the
reference's README flow; without a GPU the same binary must fail loudly (no CPU fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "psf_flow")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "psf_flow.cpp")
    libdir = os.path.join(ROOT, "tools_amd", "lib")
    if os.path.exists(BIN) and os.path.getmtime(BIN) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(libdir, "libpsf_mi355x.so"))):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", BIN, src, "-L" + libdir, "-lpsf_mi355x",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    import torch
    build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "HIP" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_flow_on_gpu():
    build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PSFPerturbation ok" in r.stdout and "PSFGPV ok" in r.stdout and "PSFGPVRing ok" in r.stdout
