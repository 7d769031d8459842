"""Bundled PPO stack with the public surface of ``rsl_rl`` v1 (the reference's [EXTERNAL]
L3 layer, absent from /root/reference and from this image): ``ActorCritic``, ``PPO``,
``RolloutStorage`` and ``OnPolicyRunner``.  Used when ``rsl_rl`` is not importable;
a user-provided ``rsl_rl`` takes precedence (``utils/task_registry.py``).

Multi-GPU (new functionality, the reference is single-process): one process per GPU
with ``torch.distributed`` (backend "nccl" = RCCL over xGMI); envs are sharded, the
policy is replicated; per PPO update there is ONE all-gather of the fused
``[returns || advantages]`` buffer (global advantage normalisation), a flattened
gradient all-reduce per mini-batch and a scalar all-reduce of the mean KL so the
adaptive learning rate stays identical on every rank.
"""
from .actor_critic import ActorCritic
from .ppo import PPO, RolloutStorage
from .runner import OnPolicyRunner



def FusedActor(*args, **kwargs):
    """Lazy import: the fused MFMA actor needs the HIP extension (GPU only)."""
    from .fused_actor import FusedActor as _F
    return _F(*args, **kwargs)


__all__ = ["ActorCritic", "PPO", "RolloutStorage", "OnPolicyRunner", "FusedActor"]
