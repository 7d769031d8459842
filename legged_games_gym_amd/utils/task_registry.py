"""name -> (task class, env cfg, train cfg) registry with the reference's API.

Mirror of reference ``legged_gym/utils/task_registry.py:44-162, 224``:
``register`` / ``get_task_class`` / ``get_cfgs`` / ``make_env`` /
``make_alg_runner`` keep their signatures, defaults, printouts-free behaviour
and error types (``ValueError`` for an unknown task or missing name).  The
game-layer ``make_dec_alg_runner`` (:164-221) is out of scope.

The PPO runner is ``rsl_rl.runners.OnPolicyRunner`` when that package is
importable, otherwise the bundled ``legged_games_gym_amd.rl.OnPolicyRunner``
(same constructor / ``learn`` / ``load`` / ``get_inference_policy`` surface).
"""
import os
from datetime import datetime

from legged_games_gym_amd import LEGGED_GYM_ROOT_DIR
from .helpers import get_args, update_cfg_from_args, class_to_dict, get_load_path, set_seed, parse_sim_params


def _runner_class():
    try:
        from rsl_rl.runners import OnPolicyRunner        # user-provided rsl_rl drops in unchanged
        return OnPolicyRunner
    except ImportError:
        from legged_games_gym_amd.rl import OnPolicyRunner
        return OnPolicyRunner


class TaskRegistry:
    def __init__(self):
        self.task_classes = {}
        self.env_cfgs = {}
        self.train_cfgs = {}
        self.experimental = {}        # task name -> caveat printed by make_env (not part of the reference's registry)

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name] = task_class
        self.env_cfgs[name] = env_cfg
        self.train_cfgs[name] = train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        train_cfg = self.train_cfgs[name]
        env_cfg = self.env_cfgs[name]
        env_cfg.seed = train_cfg.seed          # the env seeds from the *registered* train cfg (quirk Q10)
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None):
        if args is None:
            args = get_args()
        if name in self.task_classes:
            task_class = self.get_task_class(name)
        else:
            raise ValueError(f"Task with name: {name} was not registered")
        if name in self.experimental:
            import warnings
            warnings.warn(f"task '{name}' is EXPERIMENTAL in legged_games_gym_amd: {self.experimental[name]}", stacklevel=2)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        set_seed(env_cfg.seed)
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                         sim_device=args.sim_device, headless=args.headless)
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default"):
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)

        stamp = datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name
        default_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
        if log_root == "default":
            log_root = default_root
            log_dir = os.path.join(log_root, stamp)
        elif log_root is None:
            log_dir = None                   # this process writes no logs (ranks > 0 of a multi-GPU job) ...
            log_root = default_root          # ... but a resume still finds the checkpoint where rank 0 saved it
        else:
            log_dir = os.path.join(log_root, stamp)

        runner = _runner_class()(env, class_to_dict(train_cfg), log_dir, device=args.rl_device)
        if train_cfg.runner.resume:
            resume_path = self._resume_path(log_root, train_cfg)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path)             # every rank loads the same file: replicas (weights + Adam state) start identical
        return runner, train_cfg


    @staticmethod
    def _resume_path(log_root, train_cfg):
        """Checkpoint to resume from.  In a multi-process job rank 0 resolves it (before its own new run directory exists
        on disk) and the others receive the string: a rank listing the log root later could pick rank 0's fresh, empty run."""
        try:
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except ImportError:
            multi = False
        if not multi:
            return get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
        box = [None]
        if dist.get_rank() == 0:
            try:
                box[0] = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            except ValueError as exc:
                box[0] = exc
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], Exception):
            raise box[0]
        return box[0]


task_registry = TaskRegistry()
