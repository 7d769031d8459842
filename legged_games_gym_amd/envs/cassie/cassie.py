"""Cassie task (reference ``legged_gym/envs/cassie/cassie.py:42-46``): LeggedRobot plus the
``no_fly`` reward term, which is one of the terms the fused kernel evaluates (LG_REW_NO_FLY)."""
from legged_games_gym_amd.envs.base.legged_robot import LeggedRobot


class Cassie(LeggedRobot):
    pass
