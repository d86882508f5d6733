"""
Test configuration.

  -m "not gpu" : oracle vs golden vectors, host logic, C-ABI symbol table,
                 multi-process (gloo) sharding - runs anywhere, no GPU.
  -m gpu       : parity of the HIP path with the oracle - needs an MI355X.

The oracle (oracle/) is imported here and only here (plus smoke() and the
cpu_baseline leg of bench.py); the product package never imports it.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fp8-mps-metal_amd")
ORACLE = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (PKG, ORACLE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")
    config.addinivalue_line("markers", "gpu_perf: wall-clock comparisons on the GPU (dispatch regret); collected AFTER every parity test and "
                                       "report-only on a noisy box, so that `pytest -m gpu -x` always reaches the whole parity suite")


def pytest_collection_modifyitems(config, items):
    """Timing tests run last: with `-x` a wall-clock assertion that trips on a noisy box must not blank the parity tests behind it."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("gpu_perf") else 0)   # (stable: the order inside each class is kept)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import fp8_oracle
    return fp8_oracle


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle of the C oracle (built on demand with gcc)."""
    import ctypes
    import subprocess
    so = os.path.join(ORACLE, "libfp8_oracle.so")
    src = os.path.join(ORACLE, "fp8_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE, "-s"])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def native(cuda):
    """The product op layer; loading fails loudly if libfp8mi.so is missing."""
    import fp8_mi355x_lib
    fp8_mi355x_lib.load()
    import fp8_mi355x_native
    return fp8_mi355x_native


@pytest.fixture()
def patch():
    import fp8_mps_patch
    fp8_mps_patch.install()
    yield fp8_mps_patch
    fp8_mps_patch.uninstall()
