"""Test-only: loaded by the child interpreters tests/test_bench_launch.py starts (this directory is put on their PYTHONPATH).
With GF_TEST_INJECT_ORACLE=1 it installs the CPU oracle as the package's backend BEFORE bench.py's own code runs, so the
multi-rank launch / sharding / timing / reporting path of bench.py can be rehearsed on a machine without a GPU (gloo).
bench.py itself contains no such switch: without an injected backend it refuses to run without a ROCm device."""
import os
import sys

if os.environ.get("GF_TEST_INJECT_ORACLE") == "1":
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(_root, "genesis-forge_amd"))
    sys.path.insert(0, os.path.join(_root, "tests"))
    from genesis_forge_amd import _native as _nat
    from oracle_backend import OracleBackend as _OracleBackend

    _nat.set_backend(_OracleBackend(os.path.join(_root, "oracle", "libgf_oracle.so")))
