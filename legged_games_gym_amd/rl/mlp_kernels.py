"""Host side of include/legged_hip.h: lg_mlp_forward / lg_mlp_backward -- the PPO learner's MLP passes on the matrix cores.

Replaces the autograd pass over rsl_rl's ActorCritic MLPs ([EXTERNAL]) inside PPO.update() for the 48-128-64-32 shape of the
flat tasks (reference anymal_c_flat_config.py:62-65); other shapes keep the torch autograd path (``supported`` is False).
"""
import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .. import capi


def _linears(seq: nn.Sequential) -> List[nn.Linear]:
    return [m for m in seq if isinstance(m, nn.Linear)]


class MlpTrainer:
    """Forward / backward of up to two ``nn.Sequential`` MLPs (actor, critic) over a mini-batch of storage rows."""
    has_fused_minibatch = True      # lg_ppo_minibatch: forward + PPO loss + backward in one kernel (48-128-64-32 shape)

    def __init__(self, nets: Sequence[nn.Sequential], inputs: Sequence[torch.Tensor], mb: int, forward_only: bool = False):
        self.lib = capi.load_library()
        self.forward_only = forward_only           # no .grad tensors, no workspace (rollout-time users, under inference_mode)
        self.nets = list(nets)
        self.inputs = list(inputs)                 # flattened [R, in] row sources (rollout storage views)
        self.mb = int(mb)
        self.layers = [_linears(n) for n in self.nets]
        # exactly Linear-ELU-Linear-ELU-Linear-ELU-Linear with the default ELU (alpha = 1): what the kernels compute
        self.supported = all(len(l) == 4 for l in self.layers) and all(
            len(n) == 7 and all(isinstance(m, nn.Linear if i % 2 == 0 else nn.ELU) and (i % 2 == 0 or m.alpha == 1.0) for i, m in enumerate(n))
            for n in self.nets)
        if not self.supported:
            return
        dev = self.inputs[0].device
        self.outputs = [torch.empty(self.mb, l[-1].out_features, device=dev) for l in self.layers]
        self.grad_outputs = [] if forward_only else [torch.zeros_like(o) for o in self.outputs]
        self.desc = (capi.lg_mlp_net * len(self.nets))()
        self.refresh()
        need = 0 if forward_only else self.lib.lg_mlp_workspace_bytes(self.desc, len(self.nets))
        self.workspace = torch.empty(need // 4, device=dev)
        probe = self.lib.lg_mlp_forward(self.desc, len(self.nets), None, self.mb, torch.cuda.current_stream(dev).cuda_stream)
        self.supported = probe == 0                # -4: this MLP shape is not built

    def refresh(self):
        """(Re)read parameter / gradient / input addresses into the descriptors; allocates .grad where missing."""
        for n, (layers, x) in enumerate(zip(self.layers, self.inputs)):
            d = self.desc[n]
            assert x.is_contiguous() and x.dtype == torch.float32
            for i, m in enumerate(layers):
                d.weights[i], d.biases[i] = m.weight.data_ptr(), m.bias.data_ptr()
                if self.forward_only:
                    continue
                for prm in (m.weight, m.bias):
                    if prm.grad is None:
                        prm.grad = torch.zeros_like(prm)
                d.grad_weights[i], d.grad_biases[i] = m.weight.grad.data_ptr(), m.bias.grad.data_ptr()
            d.input, d.output = x.data_ptr(), self.outputs[n].data_ptr()
            if not self.forward_only:
                d.grad_output = self.grad_outputs[n].data_ptr()
            dims = [layers[0].in_features] + [m.out_features for m in layers]
            for i, v in enumerate(dims):
                d.dims[i] = v

    def key(self):
        return tuple(int(self.desc[n].weights[i] or 0) for n in range(len(self.nets)) for i in range(4)) + \
            tuple(int(self.desc[n].grad_weights[i] or 0) for n in range(len(self.nets)) for i in range(4)) + \
            tuple(int(self.desc[n].input or 0) for n in range(len(self.nets)))

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.lg_last_error().decode()}")

    def forward(self, rows: Optional[torch.Tensor]):
        """outputs[n] = nets[n](inputs[n][rows]) (no autograd graph)."""
        dev = self.inputs[0].device
        self._check(self.lib.lg_mlp_forward(self.desc, len(self.nets), rows.data_ptr() if rows is not None else None, self.mb,
                                            torch.cuda.current_stream(dev).cuda_stream), "lg_mlp_forward")
        return self.outputs

    def backward(self, rows: Optional[torch.Tensor]):
        """.grad of every Linear <- gradients for dL/d outputs = ``grad_outputs`` (overwrites, like a fresh backward)."""
        dev = self.inputs[0].device
        self._check(self.lib.lg_mlp_backward(self.desc, len(self.nets), rows.data_ptr() if rows is not None else None, self.mb,
                                             self.workspace.data_ptr(), self.workspace.numel() * 4,
                                             torch.cuda.current_stream(dev).cuda_stream), "lg_mlp_backward")

    def ppo_minibatch(self, rows: torch.Tensor, batch: "capi.lg_ppo_batch"):
        """Forward, PPO loss and backward in one pass (lg_ppo_minibatch): nets = [actor, critic]; fills every .grad, batch.d_std, batch.stats."""
        dev = self.inputs[0].device
        self._check(self.lib.lg_ppo_minibatch(self.desc, rows.data_ptr(), self.mb, C.byref(batch), self.workspace.data_ptr(),
                                              self.workspace.numel() * 4, torch.cuda.current_stream(dev).cuda_stream), "lg_ppo_minibatch")


class WideMlpTrainer(MlpTrainer):
    """The same interface for MLPs of any widths (the 512-256-128 networks of the rough tasks): lg_mlp_wide_forward (one chain
    kernel through all four layers for the compiled-in shapes, per-layer GEMMs otherwise) / lg_mlp_wide_backward (tiled dX / dW
    GEMMs, csrc/lg_gemm.h), split-bf16 or f32 MFMA arithmetic (lg_mlp_wide_set_precision).  forward() leaves the activations in
    the workspace, the backward() that follows reads them; there is no single-kernel mini-batch step for these widths
    (``has_fused_minibatch``)."""
    has_fused_minibatch = False

    def __init__(self, nets: Sequence[nn.Sequential], inputs: Sequence[torch.Tensor], mb: int, forward_only: bool = False):
        self.lib = capi.load_library()
        self.forward_only = forward_only
        self.nets, self.inputs, self.mb = list(nets), list(inputs), int(mb)
        self.layers = [_linears(n) for n in self.nets]
        self.supported = all(len(l) == 4 for l in self.layers) and all(
            len(n) == 7 and all(isinstance(m, nn.Linear if i % 2 == 0 else nn.ELU) and (i % 2 == 0 or m.alpha == 1.0) for i, m in enumerate(n))
            for n in self.nets) and all(m.out_features <= 4096 and m.in_features <= 4096 for l in self.layers for m in l)
        if not self.supported:
            return
        dev = self.inputs[0].device
        self.outputs = [torch.empty(self.mb, l[-1].out_features, device=dev) for l in self.layers]
        self.grad_outputs = [] if forward_only else [torch.zeros_like(o) for o in self.outputs]
        self.desc = (capi.lg_mlp_net * len(self.nets))()
        self.refresh()
        need = self.lib.lg_mlp_wide_workspace_bytes(self.desc, len(self.nets), self.mb)
        self.workspace = torch.empty(need // 4 + 4, device=dev)

    def _args(self, rows):
        dev = self.inputs[0].device
        return (self.desc, len(self.nets), rows.data_ptr() if rows is not None else None, self.mb, self.workspace.data_ptr(),
                self.workspace.numel() * 4, torch.cuda.current_stream(dev).cuda_stream)

    def forward(self, rows: Optional[torch.Tensor]):
        self._check(self.lib.lg_mlp_wide_forward(*self._args(rows)), "lg_mlp_wide_forward")
        return self.outputs

    def backward(self, rows: Optional[torch.Tensor]):
        self._check(self.lib.lg_mlp_wide_backward(*self._args(rows)), "lg_mlp_wide_backward")

    def ppo_minibatch(self, rows, batch):
        raise RuntimeError("no fused mini-batch kernel for the wide MLPs: forward -> lg_ppo_loss -> backward")
