from .xarm_pick_and_place import XarmPickAndPlace  # noqa: F401
