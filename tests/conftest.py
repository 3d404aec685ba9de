import fcntl
import os
import subprocess
import sys

import pytest

# PyTorch-ROCm ships its own libamdhip64 (ROCm 7.0 in this image) while librgbl_frontend.so links the system one
# (/opt/rocm, 7.2).  Whichever HIP runtime is mapped first serves the whole process; if it is the system one, torch's
# device initialisation later reports "No HIP GPUs are available".  The tests that hand torch device tensors to the
# library therefore need torch loaded first - as bench.py does.
try:
    import torch  # noqa: F401
except ImportError:  # the CPU-only parts of the suite do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def emu_lib(oracle):
    """The kernel sources compiled against the CPU SIMT emulator (test infrastructure, see tests/emu)."""
    from orb_slam3_rgbl_amd import _lib
    os.makedirs(os.path.join(TESTS, "_build"), exist_ok=True)
    with open(os.path.join(TESTS, "_build", ".lock"), "w") as lock:  # pytest-xdist workers build it once, not at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "orb_slam3_rgbl_amd", "csrc"), "emu"])
    return _lib.bind(os.path.join(TESTS, "_build", "librgbl_frontend_emu.so"))


@pytest.fixture(scope="session")
def gpu_lib(oracle):
    """The product library on a real device. Fails loudly (no fallback) if it is missing or no GPU is visible."""
    from orb_slam3_rgbl_amd import _lib
    lib = _lib.load()
    assert lib.rgbl_backend() == b"hip:gfx950"
    assert lib.rgbl_device_count() >= 1, "no HIP device visible: GPU tests must not silently pass"
    return lib
