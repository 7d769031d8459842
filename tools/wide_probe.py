#!/usr/bin/env python3
"""Time the wide-MLP learner kernels (lg_mlp_wide_forward / backward) on the reference's mini-batch shape; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r02_wide_mlp_kernel_stats.csv)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch, torch.nn as nn
from legged_games_gym_amd.rl.mlp_kernels import WideMlpTrainer

def mlp(i, o, seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(i, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, o)).cuda()

obs, mb = int(sys.argv[1]) if len(sys.argv) > 1 else 235, int(os.environ.get("WIDE_MB", "24576"))
actor, critic = mlp(obs, 12, 0), mlp(obs, 1, 1)
x = torch.randn(98304, obs, device="cuda")
rows = torch.randperm(98304, device="cuda")[:mb]
tr = WideMlpTrainer([actor, critic], [x, x], mb)
for _ in range(1 if os.environ.get("WIDE_ITERS") else 3):
    tr.forward(rows); tr.backward(rows)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("WIDE_ITERS", "20"))
for _ in range(n):
    tr.forward(rows); tr.backward(rows)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / n
flops = 6.0 * mb * sum(p.numel() for net in (actor, critic) for p in net.parameters() if p.dim() == 2)
print(f"obs {obs} mb {mb}: {ms:.3f} ms per forward+backward of both nets = {flops / ms / 1e9:.1f} TFLOP/s")
