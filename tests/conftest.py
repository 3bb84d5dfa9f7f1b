"""pytest configuration: markers and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks.
`-m gpu`       : parity of the HIP path (through the C-ABI) against the oracle.
"""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runtimes are brought up only for sessions that can run GPU tests (a GPU exists and the
    # selection is not `-m "not gpu"`).  torch (used by the full-size tests to generate panels on the
    # device) ships its own HIP runtime and must be loaded before libibdgem_hip.so pulls in the system
    # one, otherwise the second runtime of the process sees no device.  A `-m "not gpu"` session stays
    # free of any initialised runtime even on a GPU box.
    markexpr = config.getoption("markexpr", "") or ""
    if os.path.exists("/dev/kfd") and "not gpu" not in markexpr:
        try:
            import torch
            torch.cuda.is_available()
        except Exception:                           # pragma: no cover
            pass


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (the CPU checker), built on demand."""
    so = os.path.join(REPO, "oracle", "liboracle.so")
    src = os.path.join(REPO, "oracle", "ibd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    import oracle_lib
    return oracle_lib.Oracle(so)
