"""Import-path mirror of reference ``legged_gym/envs/base/legged_robot_config.py``."""
from ..configs import LeggedRobotCfg, LeggedRobotCfgPPO  # noqa: F401
