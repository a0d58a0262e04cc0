from .xarm_pick_and_place import XarmPickAndPlace  # noqa: F401
from .xarm_reach import XarmReachEnv  # noqa: F401
from .xarm_handover import XarmHandover  # noqa: F401
from .xarm_stack_tower import XarmStackTowerEnv  # noqa: F401
