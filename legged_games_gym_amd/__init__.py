"""legged_games_gym_amd -- MI355X-native legged-locomotion env hot path.

Only the path named by BASELINE.json's north_star lives here: the
``LeggedRobot.step()`` / ``post_physics_step()`` loop (reference
``legged_gym/envs/base/legged_robot.py:80-137``) behind the reference's own
``task_registry`` / ``LeggedRobotCfg`` / ``VecEnv`` surface.  The arithmetic is
in ``csrc/`` (HIP, gfx950) behind the C-ABI declared in ``include/legged_hip.h``.

The two path constants mirror ``legged_gym/__init__.py:31-34`` of the reference
so config strings such as ``"{LEGGED_GYM_ROOT_DIR}/resources/robots/..."``
keep working.  ``LEGGED_GYM_ROOT_DIR`` may be overridden with the environment
variable of the same name to point at a user's own legged_gym checkout (for
raw URDF files); the compiled model tables shipped under ``resources/models``
are used otherwise.
"""
import os

PACKAGE_DIR = os.path.dirname(os.path.realpath(__file__))
LEGGED_GYM_ROOT_DIR = os.environ.get("LEGGED_GYM_ROOT_DIR", PACKAGE_DIR)
LEGGED_GYM_ENVS_DIR = os.path.join(PACKAGE_DIR, "envs")

__all__ = ["PACKAGE_DIR", "LEGGED_GYM_ROOT_DIR", "LEGGED_GYM_ENVS_DIR"]
