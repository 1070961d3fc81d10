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


def test_host_logic_under_address_and_ub_sanitizers():
    """tools_amd/csrc/psf_host.cpp (no HIP in it) built with g++ -fsanitize=address,undefined and driven through the reference's
    invariants by tests/cpp/host_sanitize.cpp: A S_A = 0, G-digits, elimination, ring embedding, rot^-."""
    src = os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp")
    host = os.path.join(ROOT, "tools_amd", "csrc", "psf_host.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "host_sanitize")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-o", exe, src, host], capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower() or "sanitize" in b.stderr.lower()):
        pytest.skip("this g++ has no sanitizer runtime: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "HOST_SANITIZE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_oracle_under_address_and_ub_sanitizers():
    """The checker itself: oracle/*.c built with gcc -fsanitize=address,undefined and run through one small flow per scheme
    (tests/cpp/oracle_sanitize.c)."""
    exe = os.path.join(ROOT, "tests", "cpp", "oracle_sanitize")
    srcs = [os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "psf_oracle.c"),
            os.path.join(ROOT, "oracle", "psf_oracle_gpv.c")]
    b = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                        "-ffp-contract=off", "-fopenmp", "-o", exe] + srcs + ["-lm"], capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr.lower():
        pytest.skip("this gcc has no sanitizer runtime: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2"))
    assert r.returncode == 0 and "ORACLE_SANITIZE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
