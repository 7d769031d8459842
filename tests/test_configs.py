"""G2: the config surface equals the reference's own config classes, value for value.

tests/golden/configs.json was produced by executing the reference's pure-Python config files
(tools/make_golden.py); here our classes are dumped with our class_to_dict and compared."""
import json
import os

import pytest

from legged_games_gym_amd.envs import configs
from legged_games_gym_amd.envs.base.base_config import BaseConfig
from legged_games_gym_amd.utils.helpers import class_to_dict

PAIRS = {"anymal_c_rough": (configs.AnymalCRoughCfg, configs.AnymalCRoughCfgPPO),
         "anymal_c_flat": (configs.AnymalCFlatCfg, configs.AnymalCFlatCfgPPO),
         "cassie": (configs.CassieRoughCfg, configs.CassieRoughCfgPPO),
         "a1": (configs.A1RoughCfg, configs.A1RoughCfgPPO),
         "anymal_b": (configs.AnymalBRoughCfg, configs.AnymalBRoughCfgPPO),
         "base": (configs.LeggedRobotCfg, configs.LeggedRobotCfgPPO)}


def _norm(x):
    """JSON round trip (tuples -> lists) so both sides are plain data."""
    return json.loads(json.dumps(x))


@pytest.mark.parametrize("task", sorted(PAIRS))
def test_config_values_match_reference(task, golden_dir):
    with open(os.path.join(golden_dir, "configs.json")) as fh:
        gold = json.load(fh)[task]
    E, T = PAIRS[task]
    ours_env, ours_train = _norm(class_to_dict(E())), _norm(class_to_dict(T()))
    gold_env = dict(gold["env"])
    # the reference's file template differs only in the package root placeholder expansion target
    assert ours_env == gold_env
    assert ours_train == gold["train"]


def test_reward_order_is_alphabetical_and_nonzero_sets():
    # class_to_dict walks dir(): alphabetical, which fixes the reward summation order (legged_robot.py:199-203)
    flat = class_to_dict(configs.AnymalCFlatCfg().rewards.scales)
    assert list(flat) == sorted(flat)
    nz = lambda d: [k for k, v in d.items() if v != 0]
    assert nz(flat) == ["action_rate", "ang_vel_xy", "collision", "dof_acc", "feet_air_time", "lin_vel_z", "orientation",
                        "torques", "tracking_ang_vel", "tracking_lin_vel"]
    assert len(nz(class_to_dict(configs.AnymalCRoughCfg().rewards.scales))) == 9
    cas = nz(class_to_dict(configs.CassieRoughCfg().rewards.scales))
    assert len(cas) == 11 and "no_fly" in cas and "termination" in cas


def test_base_config_instantiates_nested_classes():
    a, b = configs.AnymalCFlatCfg(), configs.AnymalCFlatCfg()
    assert not isinstance(a.env, type) and not isinstance(a.rewards.scales, type)
    a.env.num_envs = 7
    assert b.env.num_envs == 4096 and configs.AnymalCFlatCfg.env.num_envs == 4096

    class C(BaseConfig):
        class inner:
            x = 1

            class deeper:
                y = 2
    c = C()
    assert c.inner.deeper.y == 2 and not isinstance(c.inner.deeper, type)


def test_derived_constants():
    cfg = configs.AnymalCFlatCfg()
    dt = cfg.control.decimation * cfg.sim.dt
    assert abs(dt - 0.02) < 1e-12
    import numpy as np
    assert int(np.ceil(cfg.env.episode_length_s / dt)) == 1000          # max_episode_length
    assert int(np.ceil(cfg.domain_rand.push_interval_s / dt)) == 750     # push_interval
    assert int(cfg.commands.resampling_time / dt) == 200                 # flat: every 200 steps
    assert int(configs.AnymalCRoughCfg().commands.resampling_time / dt) == 500
    assert len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y) == 187
    c = configs.CassieRoughCfg()
    assert len(c.terrain.measured_points_x) * len(c.terrain.measured_points_y) == 121 and c.env.num_observations == 169
