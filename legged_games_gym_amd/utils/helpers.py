"""Host-side helpers with the reference's names and semantics.

Mirrors the in-scope part of reference ``legged_gym/utils/helpers.py``:
``class_to_dict`` (:41-56, defines the *alphabetical* reward order through
``dir()``), ``update_class_from_dict`` (:58-65), ``set_seed`` (:67-77),
``parse_sim_params`` (:79-101), ``get_load_path`` (:103-125),
``update_cfg_from_args`` (:159-182) and ``get_args`` (:184-210).  The Isaac Gym
``gymutil.parse_arguments`` call is replaced by ``argparse`` exposing the same
flag names; ``export_policy_as_jit`` (:212-222, feed-forward actors); the recurrent exporter and the game-layer
loaders are out of scope.
"""
import argparse
import os
import random

import numpy as np
import torch


def class_to_dict(obj):
    """Config object -> plain dict, keys in ``dir()`` (alphabetical) order."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        if isinstance(val, list):
            out[key] = [class_to_dict(v) for v in val]
        else:
            out[key] = class_to_dict(val)
    return out


def update_class_from_dict(obj, dct):
    for key, val in dct.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    """Seeds python, numpy and torch (CPU + all GPUs); -1 draws a seed."""
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    return seed


class SimParams:
    """Plain-attribute stand-in for ``gymapi.SimParams`` (external type).

    Holds exactly the fields the env reads: ``dt``, ``substeps``, ``gravity``,
    ``up_axis``, ``use_gpu_pipeline`` and a ``physx`` namespace.
    """

    class _NS:
        pass

    def __init__(self):
        self.dt = 1.0 / 60.0
        self.substeps = 2
        self.gravity = [0.0, 0.0, -9.81]
        self.up_axis = 1
        self.use_gpu_pipeline = True
        self.physx = SimParams._NS()
        self.physx.use_gpu = True
        self.physx.num_subscenes = 0
        self.physx.num_threads = 0

    def __repr__(self):
        return f"SimParams(dt={self.dt}, substeps={self.substeps}, gravity={self.gravity})"


def parse_sim_params(args, cfg):
    """``{"sim": {...}}`` dict -> SimParams, command-line overrides applied."""
    sp = SimParams()
    sp.physx.use_gpu = getattr(args, "use_gpu", True)
    sp.physx.num_subscenes = getattr(args, "subscenes", 0)
    sp.use_gpu_pipeline = getattr(args, "use_gpu_pipeline", True)
    sim = cfg.get("sim", {}) if isinstance(cfg, dict) else {}
    for key, val in sim.items():
        if key == "physx":
            for k2, v2 in val.items():
                setattr(sp.physx, k2, v2)
        else:
            setattr(sp, key, val)
    if getattr(args, "num_threads", 0) and args.num_threads > 0:
        sp.physx.num_threads = args.num_threads
    return sp


def get_load_path(root, load_run=-1, checkpoint=-1):
    """Latest run dir (lexicographic) and highest ``model_*.pt`` (zero-padded key)."""
    try:
        runs = sorted(os.listdir(root))
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = [f for f in os.listdir(load_run) if "model" in f]
        models.sort(key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and args.num_envs is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.max_iterations is not None:
            cfg_train.runner.max_iterations = args.max_iterations
        if args.resume:
            cfg_train.runner.resume = args.resume
        for name in ("experiment_name", "run_name", "load_run", "checkpoint"):
            val = getattr(args, name)
            if val is not None:
                setattr(cfg_train.runner, name, val)
    return env_cfg, cfg_train


def export_policy_as_jit(actor_critic, path):
    """TorchScript copy of the actor for deployment: ``<path>/policy_1.pt`` (reference helpers.py:212-222; the reference's
    LSTM exporter for recurrent policies is not needed by any in-scope task)."""
    import copy
    if hasattr(actor_critic, "memory_a"):
        raise NotImplementedError("recurrent policy export is out of scope")
    os.makedirs(path, exist_ok=True)
    target = os.path.join(path, "policy_1.pt")
    actor = copy.deepcopy(actor_critic.actor).to("cpu").eval()
    torch.jit.script(actor).save(target)
    return target


def get_args(argv=None):
    """Same flags as the reference CLI (helpers.py:185-203 + the gymutil set)."""
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="anymal_c_flat")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)   # dead flag in the reference too
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    # flags gymutil.parse_arguments contributes
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--physx", action="store_true", default=True)
    args = p.parse_args(argv)
    dev = args.sim_device
    args.sim_device_type = dev.split(":")[0]
    args.compute_device_id = int(dev.split(":")[1]) if ":" in dev else 0
    args.sim_device_id = args.compute_device_id
    args.use_gpu = args.sim_device_type == "cuda"
    args.use_gpu_pipeline = args.use_gpu and args.pipeline.lower() in ("gpu", "cuda")
    args.physics_engine = "physx"
    args.device = args.sim_device_type
    return args
