"""Task registration (reference ``legged_gym/envs/__init__.py:32-56``), in-scope tasks only.

The ``a1`` / ``anymal_b`` configs and the predator-prey ``a1_game`` layer are out of scope
(SURVEY.md section 2, rows 13-16).
"""
from .base.legged_robot import LeggedRobot
from .anymal_c.anymal import Anymal
from .cassie.cassie import Cassie
from .configs import (LeggedRobotCfg, LeggedRobotCfgPPO, AnymalCRoughCfg, AnymalCRoughCfgPPO, AnymalCFlatCfg,
                      AnymalCFlatCfgPPO, CassieRoughCfg, CassieRoughCfgPPO)
from legged_games_gym_amd.utils.task_registry import task_registry

task_registry.register("anymal_c_rough", Anymal, AnymalCRoughCfg(), AnymalCRoughCfgPPO())
task_registry.register("anymal_c_flat", Anymal, AnymalCFlatCfg(), AnymalCFlatCfgPPO())
task_registry.register("cassie", Cassie, CassieRoughCfg(), CassieRoughCfgPPO())
