"""Task registration (reference ``legged_gym/envs/__init__.py:32-56``), in-scope tasks only.

The predator-prey ``a1_game`` layer (low/high-level game tasks) is out of scope.
"""
from .base.legged_robot import LeggedRobot
from .anymal_c.anymal import Anymal
from .cassie.cassie import Cassie
from .configs import (LeggedRobotCfg, LeggedRobotCfgPPO, AnymalCRoughCfg, AnymalCRoughCfgPPO, AnymalCFlatCfg,
                      AnymalCFlatCfgPPO, AnymalBRoughCfg, AnymalBRoughCfgPPO, A1RoughCfg, A1RoughCfgPPO,
                      CassieRoughCfg, CassieRoughCfgPPO)
from legged_games_gym_amd.utils.task_registry import task_registry

task_registry.register("anymal_c_rough", Anymal, AnymalCRoughCfg(), AnymalCRoughCfgPPO())
task_registry.register("anymal_c_flat", Anymal, AnymalCFlatCfg(), AnymalCFlatCfgPPO())
task_registry.register("anymal_b", Anymal, AnymalBRoughCfg(), AnymalBRoughCfgPPO())
task_registry.register("a1", LeggedRobot, A1RoughCfg(), A1RoughCfgPPO())
task_registry.register("cassie", Cassie, CassieRoughCfg(), CassieRoughCfgPPO())
