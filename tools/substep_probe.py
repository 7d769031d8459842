import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
from tests.test_gpu_parity import pair, init_both, put, get
task = sys.argv[1] if len(sys.argv) > 1 else "anymal_c_flat"
N = 512
cfg, robot, p, names, o, d = pair(task, N)
init_both(o, d, N)
for k in ("root_states", "dof_state", "commands"):
    a, b = o.buf[k], get(d, k)
    i = np.unravel_index(np.abs(a - b).argmax(), a.shape)
    print("reset", k, np.abs(a - b).max(), i, a[i], b[i])
rng = np.random.default_rng(2)
root = o.buf["root_states"].copy()
root[:, 2] = rng.uniform(0.35, 0.75, N) if task != "cassie" else rng.uniform(0.7, 1.1, N)
quat = np.array([0, 0, 0, 1.0]) + rng.normal(0, 0.15, (N, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
root[:, 3:7] = quat; root[:, 7:13] = rng.normal(0, 0.7, (N, 6))
put(o, d, "root_states", root)
dof = o.buf["dof_state"].copy(); dof[:, 1] = rng.normal(0, 2.0, dof.shape[0]); put(o, d, "dof_state", dof)
qd0 = dof.reshape(N, 12, 2)[..., 1].copy()
tau = rng.normal(0, 20.0, (N, 12)).astype(np.float32)
o.physics_substep(tau, True); d.physics_substep(torch.from_numpy(tau), True)
q_o, q_d = o.buf["dof_state"].reshape(N, 12, 2), get(d, "dof_state").reshape(N, 12, 2)
dv = np.abs(q_o[..., 1] - q_d[..., 1]).max(axis=1)
acc = np.abs(q_o[..., 1] - qd0).max(axis=1) / 0.005
fmax = np.abs(o.buf["contact_forces"]).max(axis=(1, 2))
order = np.argsort(-dv)
print("worst envs: dv, |dqd|/dt (rad/s^2), max contact force")
for i in order[:12]: print(i, "%.2e" % dv[i], "%.1f" % acc[i], "%.1f" % fmax[i])
nc = fmax < 1e-9
print("no-contact envs:", nc.sum(), "max dv %.2e" % dv[nc].max(), "median dv all %.2e" % np.median(dv), "rel err max %.2e" % (dv / (acc * 0.005 + 1e-3)).max())
print("contact envs with F<2000:", ((fmax > 0) & (fmax < 2000)).sum(), "max dv %.2e" % dv[(fmax > 0) & (fmax < 2000)].max())
