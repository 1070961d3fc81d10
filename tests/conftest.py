import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


EXP_LIB = os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x_exp.so")


def exp_env(**switches):
    """Environment of a subprocess that runs the EXPERIMENTS build of the library (tools_amd/csrc `make exp`, -DPSF_EXPERIMENTS) with the given PSF_* switches: the
    release library reads no such switch.  Without switches the release library is what runs -- so a comparison "switch X == default" is also a comparison of the
    two builds."""
    env = {k: v for k, v in os.environ.items() if not (k.startswith("PSF_") and k not in ("PSF_SYSTEM_HIP",))}
    if switches:
        if not os.path.exists(EXP_LIB):
            pytest.skip("the experiments build of the library is missing (make -C tools_amd/csrc exp)")
        env["PSF_LIB"] = EXP_LIB
        env.update({k: str(v) for k, v in switches.items()})
    return env


@pytest.fixture
def exp_lib(monkeypatch):
    """In-process tests of the experiment switches: for the duration of the test tools_amd talks to the experiments build (same ABI, same sources; the PSF_* switches
    the test sets with monkeypatch.setenv are alive there and dead in the release library).  Handles keep the library they were created with."""
    if not os.path.exists(EXP_LIB):
        pytest.skip("the experiments build of the library is missing (make -C tools_amd/csrc exp)")
    from tools_amd import _ffi
    _ffi.lib()                                   # the release library first: one HIP runtime, loaded in the usual order
    L = _ffi.open_library(EXP_LIB)
    monkeypatch.setattr(_ffi, "_lib", L)
    yield L


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, as __graft_entry__.build() does."""
    lib = os.path.join(ROOT, "tools_amd", "lib", "libpsf_mi355x.so")
    orc = os.path.join(ROOT, "oracle", "libpsf_oracle.so")
    if os.path.exists(lib) and os.path.exists(orc) and os.path.exists(EXP_LIB):
        return
    import subprocess
    try:
        if not os.path.exists(orc):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        if not os.path.exists(lib) or not os.path.exists(EXP_LIB):
            subprocess.check_call(["make", "-j4", "-C", os.path.join(ROOT, "tools_amd", "csrc"), "all", "exp"], stdout=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError) as exc:      # a host without ROCm still runs the oracle-only tests
        sys.stderr.write(f"[conftest] native build failed ({exc}); tests that need the library will fail or skip\n")


def _have_gpu():
    """True when a HIP device is usable: /dev/kfd present and the product library finds a device."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import ctypes
        from tools_amd import _ffi
        name = ctypes.create_string_buffer(64)
        cus = ctypes.c_int(0)
        return _ffi.lib().psf_device_info(0, name, 64, ctypes.byref(cus)) == 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) on a host without a HIP device; a per-test ceiling (pytest-timeout,
    when installed) so that one stalled test cannot hold the whole suite."""
    if any(item.get_closest_marker("gpu") for item in items) and not _have_gpu():
        skip = pytest.mark.skip(reason="no HIP device on this host (run with -m gpu on the MI355X box)")
        for item in items:
            if item.get_closest_marker("gpu"):
                item.add_marker(skip)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "ref_kats.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
