"""The reference at the sizes that get timed.

Every other reference-derived fixture holds at most 130 envs; above that the comparisons were HIP against the oracle, both
behind the same host logic.  Two legs close the chain (missing #4):

* container only (``/root/reference`` present): the reference's own ``ManagedEnvironment.step`` (managed_env.py:274-334, imported
  under the stubs of tools/ref_stubs.py) at 4 096 and 65 536 envs for 20 steps against this package on the CPU oracle — the same
  actions, the same Philox draws, EVERY env of every step, no fixture file in between (``tools/gen_golden.py check_at_size``, run
  in a child process: the stubs patch ``torch.Tensor.uniform_`` and install fake ``genesis`` modules);
* everywhere: compact fixtures recorded from the same reference runs (``tools/gen_golden.py at_size``; per-step mask popcounts,
  f64 sums and sums of squares of reward / observation / command over ALL envs, integer sums of the episode counters, the logged
  scalars, and all outputs of a strided 64-env sample) against the oracle (CPU) and against the HIP kernels (``-m gpu``) at
  4 096 and 65 536 envs — and at both sizes WITH the contacts example's ContactManagers (``atsize_go2c_*``: contact kernel, air time,
  contact-force termination / rewards / observation; the 65 536-env reference run took 50 minutes of the serial Taichi emulation,
  ``tools/gen_golden.py at_size go2c_65536``).
"""
import os
import subprocess
import sys

import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REFERENCE = os.path.isdir("/root/reference/genesis_forge")


@pytest.mark.skipif(not HAVE_REFERENCE, reason="the reference only exists in the build container")
@pytest.mark.parametrize("n", [4096, 65536])
def test_reference_itself_equals_package_at_size(n, oracle_lib_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "check_at_size", str(n)],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, GF_DEVICE="cpu"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"reference == package (oracle backend) at {n} envs" in p.stdout


@pytest.mark.skipif(not HAVE_REFERENCE, reason="the reference only exists in the build container")
def test_reference_itself_equals_package_with_contact_managers(oracle_lib_path):
    """The same lockstep check with the contacts example's two ContactManagers (contact-force termination, feet-air-time and
    has-contact rewards, a contact-force observation item): the reference's Taichi kernel SOURCE runs under the serial emulation
    of tools/ref_stubs.py, ≈ 2 ms per env and step — 512 envs (8 tiles) here, 4 096 in the fixture below."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "check_at_size", "512", "contacts"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, GF_DEVICE="cpu"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "reference == package (oracle backend) at 512 envs with contact managers" in p.stdout


FIXTURES = ["atsize_go2_4096", "atsize_go2_65536", "atsize_go2c_4096", "atsize_go2c_65536",   # (…c: with the contact managers)
            "atsize_rough_16384"]   # BASELINE config 3's structure at its size: terrain spawn / height lookups, two ContactManagers


@pytest.mark.parametrize("name", FIXTURES)
def test_at_size_fixture_oracle(oracle_backend, name):
    fix = helpers.load(name)
    got, logs = helpers.replay_at_size(fix, dev="cpu")
    helpers.compare_at_size(fix, got, logs)
    assert int(fix["terminated_count"].sum() + fix["truncated_count"].sum()) > int(fix["n"]) // 10
    assert bool(fix["contacts"]) == name.startswith("atsize_go2c")


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_at_size_fixture_hip(hip_backend, name):
    fix = helpers.load(name)
    got, logs = helpers.replay_at_size(fix, dev="cuda")
    helpers.compare_at_size(fix, got, logs)
