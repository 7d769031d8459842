"""Rollout-time actor on the matrix cores: ``actions = actor(obs) + std * eps`` in ONE kernel
(``lg_policy_act``, csrc/lg_policy.h) instead of ~10 torch launches.

Wraps an ``ActorCritic`` (bundled or rsl_rl's: anything with ``.actor`` = Sequential of Linear/ELU and ``.std``).
The torch module stays the owner of the parameters; call ``sync()`` after an optimiser step to re-upload them.
Numerics: fp32 MFMA (exact fp32 FMA chains); tests compare against the torch fp32 forward.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import capi


class FusedActor:
    def __init__(self, actor_critic, device, seed: int = 1, step_counter: torch.Tensor = None):
        self.lib = capi.load_library()
        self.ac = actor_critic
        self.device = torch.device(device)
        self.seed = int(seed)
        self.step_counter = step_counter          # device int64[1] shared with the env (graph replay), or None
        self._host_step = 0
        self.handle = C.c_void_p()
        self._out = None
        self.sync()

    def _layers(self):
        mods = list(self.ac.actor)
        lin = [m for m in mods if isinstance(m, nn.Linear)]
        act = [m for m in mods if not isinstance(m, nn.Linear)]
        if len(lin) != 4 or not all(isinstance(m, nn.ELU) for m in act):
            raise ValueError("FusedActor supports 3 hidden layers with ELU (the reference's policy configs)")
        return lin

    def sync(self):
        """(Re-)upload the actor's weights; call after every optimiser step."""
        lin = self._layers()
        dims = (C.c_int32 * 5)(lin[0].in_features, lin[0].out_features, lin[1].out_features, lin[2].out_features, lin[3].out_features)
        ws = [np.ascontiguousarray(m.weight.detach().float().cpu().numpy()) for m in lin]
        bs = [np.ascontiguousarray(m.bias.detach().float().cpu().numpy()) for m in lin]
        std = np.ascontiguousarray(self.ac.std.detach().float().cpu().numpy())
        PF = C.POINTER(C.c_float)
        wp = (PF * 4)(*[w.ctypes.data_as(PF) for w in ws])
        bp = (PF * 4)(*[b.ctypes.data_as(PF) for b in bs])
        if self.handle:
            self.lib.lg_policy_destroy(self.handle)
            self.handle = C.c_void_p()
        rc = self.lib.lg_policy_create(dims, wp, bp, std.ctypes.data_as(PF), self.device.index or 0, C.byref(self.handle))
        if rc != 0:
            raise RuntimeError(f"lg_policy_create failed ({rc}): {self.lib.lg_last_error().decode()}")
        self.num_actions = lin[3].out_features

    def sync_device(self):
        """Refresh the kernel's weights from the torch parameters ON THE DEVICE (``lg_policy_load_device``): no host copy, no
        synchronisation -- cheap enough to call after every PPO update."""
        lin = self._layers()
        ptr = C.c_void_p * 4
        ws = ptr(*[m.weight.data_ptr() for m in lin]); bs = ptr(*[m.bias.data_ptr() for m in lin])
        for m in lin:
            if not (m.weight.is_contiguous() and m.weight.dtype == torch.float32 and m.weight.is_cuda):
                raise ValueError("FusedActor.sync_device needs contiguous float32 CUDA parameters")
        rc = self.lib.lg_policy_load_device(self.handle, ws, bs, self.ac.std.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"lg_policy_load_device failed ({rc}): {self.lib.lg_last_error().decode()}")

    def output_buffers(self, n):
        """(actions, mean) tensors the kernels write into (re-used between calls; clone to keep a value)."""
        if self._out is None or self._out[0].shape[0] != n:
            self._out = (torch.empty(n, self.num_actions, device=self.device), torch.empty(n, self.num_actions, device=self.device))
        return self._out

    def _call(self, obs, deterministic, want_mean):
        obs = obs if (obs.dtype == torch.float32 and obs.is_contiguous()) else obs.float().contiguous()
        n = obs.shape[0]
        actions, mean = self.output_buffers(n)
        if self.step_counter is not None:
            step, ctr = -1, self.step_counter.data_ptr()
        else:
            self._host_step += 1
            step, ctr = self._host_step, None
        rc = self.lib.lg_policy_act(self.handle, obs.data_ptr(), actions.data_ptr(), mean.data_ptr() if want_mean else None, n,
                                    self.seed, step, ctr, int(deterministic), torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"lg_policy_act failed ({rc}): {self.lib.lg_last_error().decode()}")
        return actions, mean

    def act(self, obs):
        """Sampled actions (ActorCritic.act)."""
        return self._call(obs, False, False)[0]

    def act_with_mean(self, obs):
        return self._call(obs, False, True)

    def act_inference(self, obs):
        return self._call(obs, True, False)[0]

    def __del__(self):
        try:
            if self.handle:
                self.lib.lg_policy_destroy(self.handle)
        except Exception:
            pass
