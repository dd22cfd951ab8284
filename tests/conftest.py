"""pytest wiring: import path of the package, the ``gpu`` marker, and the two backends.

* ``-m "not gpu"`` tests run the host logic against the CPU ORACLE (oracle/libgf_oracle.so) injected
  as the backend — the package itself never loads the oracle.
* ``-m gpu`` tests run the HIP library (libgf_step.so) on cuda:0 and compare it with the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu on the GPU box")


def _ensure_oracle():
    so = os.path.join(ROOT, "oracle", "libgf_oracle.so")
    src = os.path.join(ROOT, "oracle", "gf_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


@pytest.fixture(scope="session")
def oracle_lib_path():
    return _ensure_oracle()


@pytest.fixture()
def oracle_backend(oracle_lib_path):
    """Run the package's host logic on CPU tensors with the oracle as the compute backend."""
    import torch
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    old_dev = gs.device
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    yield nat.get_backend()
    nat.set_backend(None)
    gs.device = old_dev


@pytest.fixture()
def hip_backend():
    import torch
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    gs.set_device("cuda:0")
    nat.set_backend(None)
    return nat.get_backend()
