from .base import BaseManager
from .reward_manager import RewardManager
from .termination_manager import TerminationManager
from .action import BaseActionManager, PositionActionManager, PositionWithinLimitsActionManager
from .command import CommandManager, VelocityCommandManager
from .gait_command import GaitCommandManager
from .contact import ContactManager
from .terrain_manager import TerrainManager
from .entity_manager import EntityManager
from .observation_manager import ObservationManager
from .config import MdpFnClass, ResetMdpFnClass

__all__ = [
    "BaseManager", "RewardManager", "TerminationManager", "CommandManager", "VelocityCommandManager", "GaitCommandManager",
    "BaseActionManager", "PositionActionManager", "PositionWithinLimitsActionManager", "ContactManager",
    "TerrainManager", "EntityManager", "ObservationManager", "MdpFnClass", "ResetMdpFnClass",
]
