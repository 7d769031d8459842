"""Manual smoke run, the counterpart of the reference's only test (legged_gym/tests/test_env.py:42-56): at most 10 envs of the
chosen task, zero actions, ten episodes' worth of policy steps, then "Done".  Here it also checks what the reference leaves
to the eye: every returned buffer stays finite and episodes end by time-out or fall, never by a numerical blow-up.

    python -m legged_games_gym_amd.tests.test_env --task=anymal_c_flat
"""
import torch

from legged_games_gym_amd.envs import task_registry
from legged_games_gym_amd.utils import get_args


def test_env(args, steps=None):
    env_cfg, _ = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 10)
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    steps = int(10 * env.max_episode_length) if steps is None else steps
    zero = torch.zeros(env.num_envs, env.num_actions, device=env.device)
    resets = time_outs = 0
    for _ in range(steps):
        obs, priv, rew, done, info = env.step(zero)
        resets += int(done.sum())
        time_outs += int(info["time_outs"].sum()) if "time_outs" in info else 0
    ok = bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all()) and bool(torch.isfinite(env.root_states).all())
    print(f"{steps} steps x {env.num_envs} envs: {resets} resets ({time_outs} time-outs), state finite: {ok}")
    if not ok:
        raise RuntimeError("non-finite state")
    print("Done")
    return resets, time_outs


test_env.__test__ = False          # a script entry point with the reference's name, not a pytest case


if __name__ == "__main__":
    test_env(get_args())
