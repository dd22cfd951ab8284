"""End-to-end parity with the REFERENCE: golden trajectories recorded from /root/reference's own
ManagedEnvironment.step (tools/gen_golden.py) are replayed through this package's host logic.
CPU run: oracle as compute backend (pins the oracle + the host orchestration: phase order, extras keys,
quirks q1-q9).  GPU run: the HIP kernels must reproduce the same fixtures."""
import pytest

import helpers


@pytest.mark.parametrize("name", ["traj_go2_cmd", "traj_go2_contacts_hist", "traj_go2_rough"])
def test_trajectory_matches_reference_cpu_oracle(oracle_backend, name):
    fix = helpers.load(name)
    res = helpers.replay_trajectory(fix, "cpu")
    helpers.compare_trajectory(fix, res)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["traj_go2_cmd", "traj_go2_contacts_hist", "traj_go2_rough"])
def test_trajectory_matches_reference_hip(hip_backend, name):
    fix = helpers.load(name)
    res = helpers.replay_trajectory(fix, "cuda")
    helpers.compare_trajectory(fix, res)
