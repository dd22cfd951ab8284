"""The task configs moved into the package (genesis_forge_amd/tasks.py) so that bench.py and tools/ do not import from tests/;
the tests keep importing them under their old module name."""
from genesis_forge_amd.tasks import *  # noqa: F401,F403
from genesis_forge_amd.tasks import _std_obs, _gait_reset_with_curriculum  # noqa: F401
