"""Torch-owned state buffers bound to one ``lg_sim`` handle.

The reference lets Isaac Gym own ``root_states`` / ``dof_state`` /
``contact_forces`` and wraps them zero-copy (``gymtorch.wrap_tensor``,
``legged_robot.py:523-529``).  Here the direction is inverted: torch allocates
every buffer (``packing.buffer_spec``) and the HIP library receives the raw
device pointers through ``lg_bind``.  PyTorch is plumbing only: memory and
the current HIP stream.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import capi
from .utils import packing

_TORCH_DTYPES = {"float32": torch.float32, "bool": torch.bool, "int64": torch.int64, "int32": torch.int32,
                 "int16": torch.int16}


class DeviceSim:
    def __init__(self, params: capi.lg_params, model: capi.lg_robot_model, robot, device: torch.device,
                 actuator_weights: Optional[np.ndarray] = None, height_samples: Optional[np.ndarray] = None,
                 terrain_origins: Optional[np.ndarray] = None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                f"the legged-robot hot path runs on an AMD GPU only (got device {device}); there is no CPU product path")
        self.device = device
        self.params, self.model, self.robot = params, model, robot
        self.sim = capi.Sim(params, model, actuator_weights, device.index or 0)
        self.buf: Dict[str, torch.Tensor] = {}
        for name, (shape, dt) in packing.buffer_spec(params, robot).items():
            self.buf[name] = torch.zeros(shape, dtype=_TORCH_DTYPES[dt], device=device)
        self.buf["friction_coeffs"].fill_(1.0)
        if height_samples is not None:
            self.set_terrain(height_samples, terrain_origins)
        elif params.terrain_type != capi.TERRAIN_HEIGHTFIELD:
            self.rebind()
        # a height-field sim is bound by the set_terrain() call that must follow

    def set_terrain(self, height_samples: np.ndarray, terrain_origins: np.ndarray):
        self.buf["height_samples"] = torch.from_numpy(np.ascontiguousarray(height_samples, dtype=np.int16)).to(self.device)
        self.buf["terrain_origins"] = torch.from_numpy(np.ascontiguousarray(terrain_origins, dtype=np.float32)).to(self.device)
        self.rebind()

    def rebind(self):
        self.sim.bind({k: v.data_ptr() for k, v in self.buf.items()})

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def set_obs_output(self, obs: torch.Tensor):
        """Where the next step writes its observations (ping-pong, see LeggedRobot.step)."""
        self._obs_out = obs
        self.sim.set_obs_buffer(obs.data_ptr())

    def set_deferred_extras(self, on: bool):
        """``lg_set_deferred_extras``: steps leave extras["episode"] to the next step's launch (rollout graphs); see flush_extras."""
        self.sim.set_deferred_extras(on)

    def flush_extras(self, counter: int = -1):
        """``lg_extras_flush``: publish what the last deferred step left (capturable: the last node of a rollout graph)."""
        self.sim.extras_flush(counter, self._stream())

    def step(self, actions: torch.Tensor, counter: int):
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        n = self.robot.num_dof
        if actions.shape != (self.params.num_envs, n):
            raise ValueError(f"actions must be [{self.params.num_envs},{n}], got {tuple(actions.shape)}")
        self._keep = actions
        self.sim.step(actions.data_ptr(), counter, self._stream())

    def step_policy(self, fused_actor, obs: torch.Tensor, counter: int, deterministic: bool = False, want_mean: bool = True):
        """``actions = actor(obs) + std * eps`` and the env step in ONE launch (``lg_step_policy``); returns the
        (actions, mean) tensors owned by ``fused_actor``.  Raises if the sim / actor pair is not a compiled fused shape."""
        if obs.dtype != torch.float32 or not obs.is_contiguous() or obs.device != self.device:
            raise ValueError("obs must be a contiguous float32 tensor on the sim device")
        # obs may be the sim's own output buffer: a workgroup reads only its 16 envs' rows, and before it writes them
        actions, mean = fused_actor.output_buffers(self.params.num_envs)
        self.sim.step_policy(fused_actor.handle, obs.data_ptr(), actions.data_ptr(), mean.data_ptr() if want_mean else None,
                             fused_actor.seed, deterministic, counter, self._stream())
        return actions, mean

    def rollout_policy(self, fused_actor, storage: Dict[str, torch.Tensor], counter: int, deterministic: bool = False, obs0: Optional[torch.Tensor] = None):
        """``lg_rollout_policy``: ``steps`` fused policy steps in ONE launch.  ``storage``: contiguous float32 ``obs`` [T+1, N, num_obs]
        (``obs[0]`` = the current observations), ``actions`` / optional ``mean`` [T, N, num_actions], ``rew`` [T, N], and bool / uint8
        ``dones`` / ``time_outs`` [T, N], all on the sim device."""
        T, N, n = storage["actions"].shape[0], self.params.num_envs, self.robot.num_dof
        want = {"obs": (T + 1, N, self.params.num_obs), "actions": (T, N, n), "rew": (T, N), "dones": (T, N), "time_outs": (T, N)}
        if "mean" in storage and storage["mean"] is not None:
            want["mean"] = (T, N, n)
        for k, shape in want.items():
            t = storage[k]
            ok_dtype = t.dtype in (torch.bool, torch.uint8) if k in ("dones", "time_outs") else t.dtype == torch.float32
            if tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device or not ok_dtype:
                raise ValueError(f"rollout storage '{k}' must be a contiguous {shape} tensor on {self.device} (got {tuple(t.shape)}, {t.dtype}, {t.device})")
        mean = storage.get("mean")
        if obs0 is not None and (tuple(obs0.shape) != (N, self.params.num_obs) or not obs0.is_contiguous() or obs0.dtype != torch.float32 or obs0.device != self.device):
            raise ValueError("obs0 must be a contiguous float32 [num_envs, num_obs] tensor on the sim device")
        self._keep_roll = (storage, obs0)
        self.sim.rollout_policy(fused_actor.handle, T, storage["obs"].data_ptr(), storage["actions"].data_ptr(), mean.data_ptr() if mean is not None else None,
                                storage["rew"].data_ptr(), storage["dones"].data_ptr(), storage["time_outs"].data_ptr(), fused_actor.seed, deterministic,
                                counter, self._stream(), obs0.data_ptr() if obs0 is not None else None)

    def reset_idx(self, env_ids: torch.Tensor, counter: int):
        ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
        if ids.numel() == 0:
            return
        self._keep_ids = ids
        self.sim.reset_idx(ids.data_ptr(), ids.numel(), counter, self._stream())

    def physics_substep(self, torques: torch.Tensor, write_contacts: bool = True):
        t = torques.to(device=self.device, dtype=torch.float32).contiguous()
        self._keep = t
        self.sim.physics_substep(t.data_ptr(), int(write_contacts), self._stream())

    def actuator_forward(self, pos_err, vel, hidden, cell):
        pe = pos_err.to(self.device, torch.float32).contiguous().view(-1)
        ve = vel.to(self.device, torch.float32).contiguous().view(-1)
        out = torch.empty_like(pe)
        self.sim.actuator_forward(pe.data_ptr(), ve.data_ptr(), out.data_ptr(), hidden.data_ptr(), cell.data_ptr(),
                                  pe.numel(), self._stream())
        return out

    def compute_observations_only(self, counter: int):
        self.sim.compute_observations_only(counter, self._stream())

    def close(self):
        self.sim.close()
