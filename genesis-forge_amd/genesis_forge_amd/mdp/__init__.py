from . import reset
from . import rewards
from . import terminations
from . import observations

__all__ = ["rewards", "terminations", "observations", "reset"]
