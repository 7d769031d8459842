#!/usr/bin/env python3
"""TRACE=1 needs the -DLG_PROFILE build: python tools/profile_sections.py build, then LG_HIP_LIB=<csrc>/liblegged_hip_prof.so.
Times the learner kernels (lg_mlp_forward / lg_mlp_backward) against torch autograd on the flat actor+critic shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch, torch.nn as nn
from legged_games_gym_amd.rl.mlp_kernels import MlpTrainer

def mlp(i, o):
    return nn.Sequential(nn.Linear(i, 128), nn.ELU(), nn.Linear(128, 64), nn.ELU(), nn.Linear(64, 32), nn.ELU(), nn.Linear(32, o)).cuda()

mb, R = int(os.environ.get("MB", 24576)), 98304
actor, critic = mlp(48, 12), mlp(48, 1)
x = torch.randn(R, 48, device="cuda")
rows = torch.randperm(R, device="cuda")[:mb]
tr = MlpTrainer([actor, critic], [x, x], mb)
assert tr.supported

def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def torch_step():
    xb = x[rows]
    mu, v = actor(xb), critic(xb)
    torch.autograd.backward([mu, v], [tr.grad_outputs[0], tr.grad_outputs[1]])

flops_fwd = 2 * mb * sum(a * b for net in (actor, critic) for a, b in [(m.in_features, m.out_features) for m in net if isinstance(m, nn.Linear)])
tf, tb, tt = timeit(lambda: tr.forward(rows)), timeit(lambda: tr.backward(rows)), timeit(torch_step, 50)
print(f"mb {mb}: lg_mlp_forward {tf:.1f} us ({flops_fwd / tf / 1e6:.1f} TFLOP/s)  lg_mlp_backward(+recompute, reduce) {tb:.1f} us "
      f"({3 * flops_fwd / tb / 1e6:.1f} TFLOP/s)  torch fwd+bwd {tt:.1f} us")

if os.environ.get("TRACE"):
    import ctypes as C
    lib = tr.lib
    for name, fn in (("forward", lambda: tr.forward(rows)), ("backward", lambda: tr.backward(rows))):
        buf = torch.zeros(64 + 2 * 1024, dtype=torch.int64, device="cuda")
        lib.lg_mlp_trace(buf.data_ptr()); fn(); torch.cuda.synchronize(); lib.lg_mlp_trace(None)
        t = buf.cpu().tolist()[:60]; t = [v for v in t if v]
        print(name, "stamps:", len(t), "deltas (s_memtime ticks):", [t[i + 1] - t[i] for i in range(len(t) - 1)], "total", t[-1] - t[0])

# fused mini-batch kernel (forward + PPO loss + backward): timing and phase trace
from legged_games_gym_amd import capi
A = 12
stor = {k: torch.randn(R, d, device="cuda") for k, d in (("actions", A), ("old_mu", A), ("old_lp", 1), ("adv", 1), ("old_v", 1), ("ret", 1))}
stor["old_sigma"] = torch.full((R, A), 0.8, device="cuda")
std, d_std, stats = torch.full((A,), 0.8, device="cuda"), torch.zeros(A, device="cuda"), torch.zeros(4, device="cuda")
b = capi.lg_ppo_batch()
b.actions, b.old_log_prob, b.old_mu, b.old_sigma = (stor[k].data_ptr() for k in ("actions", "old_lp", "old_mu", "old_sigma"))
b.advantages, b.old_values, b.returns, b.std = stor["adv"].data_ptr(), stor["old_v"].data_ptr(), stor["ret"].data_ptr(), std.data_ptr()
b.clip, b.value_coef, b.entropy_coef, b.use_clipped_value, b.d_std, b.stats = 0.2, 1.0, 0.01, 1, d_std.data_ptr(), stats.data_ptr()
tm = timeit(lambda: tr.ppo_minibatch(rows, b))
print(f"mb {mb}: lg_ppo_minibatch (forward + loss + backward + reduce) {tm:.1f} us ({3 * flops_fwd / tm / 1e6:.1f} TFLOP/s)")
if os.environ.get("TRACE"):
    buf = torch.zeros(64 + 2 * 1024, dtype=torch.int64, device="cuda")
    tr.lib.lg_mlp_trace(buf.data_ptr()); tr.ppo_minibatch(rows, b); torch.cuda.synchronize(); tr.lib.lg_mlp_trace(None)
    raw = buf.cpu().tolist()
    t = [v for v in raw[:60] if v]
    print("ppo_minibatch stamps:", len(t), "deltas:", [t[i + 1] - t[i] for i in range(len(t) - 1)], "total", t[-1] - t[0])
    if not any(raw[64:]):
        print("(per-workgroup clocks: -DLG_PROFILE build only)"); sys.exit(0)
    print(f"workgroup (0,0) on the wall clock: row-tile loop ends {(raw[62] - raw[61]) / 100.0:.2f} us after its start, kernel body ends at {(raw[63] - raw[61]) / 100.0:.2f} us")
    import numpy as np
    wg = np.array(raw[64:], dtype=np.int64).reshape(-1, 2)
    wg = wg[wg[:, 0] > 0]
    t0 = wg[:, 0].min()
    st, en = (wg[:, 0] - t0) / 100.0, (wg[:, 1] - t0) / 100.0            # 100 MHz wall clock -> us
    print(f"workgroups {len(wg)}: start us median {np.median(st):.2f} max {st.max():.2f};  end us min {en.min():.2f} median {np.median(en):.2f} "
          f"p90 {np.percentile(en, 90):.2f} max {en.max():.2f};  own duration median {np.median(en - st):.2f} max {(en - st).max():.2f}")
    h = len(wg) // 2
    print(f"  actor half: end median {np.median(en[:h]):.2f} max {en[:h].max():.2f};  critic half: end median {np.median(en[h:]):.2f} max {en[h:].max():.2f}")
    order = np.argsort(en)[-8:]
    print("  last to finish (wg: start, end):", [(int(i), round(float(st[i]), 2), round(float(en[i]), 2)) for i in order])
