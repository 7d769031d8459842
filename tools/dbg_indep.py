import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from common import make_setup, grid_origins, randomize_env_params
from legged_games_gym_amd.device_sim import DeviceSim
def get(d, k): return d.buf[k].detach().cpu().numpy()
sims = []
for N in (4096, 1000):
    cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N)
    d = DeviceSim(p, model, robot, torch.device("cuda:0"), w)
    fr, dm = randomize_env_params(4096, 5)
    d.buf["env_origins"].copy_(torch.from_numpy(grid_origins(4096)[:N]))
    d.buf["friction_coeffs"].copy_(torch.from_numpy(fr[:N])); d.buf["base_mass_delta"].copy_(torch.from_numpy(dm[:N]))
    d.reset_idx(torch.arange(N, dtype=torch.int32), 0)
    sims.append(d)
g = torch.Generator(device="cuda").manual_seed(3)
acts = torch.randn(40, 4096, 12, device="cuda", generator=g)
for it in range(40):
    for d, N in zip(sims, (4096, 1000)):
        d.step(acts[it, :N].contiguous(), it + 1)
    a, c = sims
    for k in ("root_states", "dof_state", "obs_buf", "rew_buf", "reset_buf", "sea_hidden_state", "sea_cell_state", "commands", "last_actions", "feet_air_time"):
        x, y = get(a, k), get(c, k)
        if k.startswith("sea_"):
            x = x.reshape(2, 4096, -1)[:, :1000]; y = y.reshape(2, 1000, -1)
            bad = np.nonzero((x != y).any(axis=(0, 2)))[0]
        else:
            x = x.reshape(4096, -1)[:1000]; y = y.reshape(1000, -1)
            bad = np.nonzero((x != y).any(axis=1))[0]
        if len(bad):
            print("step", it + 1, k, "envs differing:", len(bad), bad[:10], "resets there (4096 run):", get(a, "reset_buf")[bad[:10]])
    if any((get(a, k).reshape(4096, -1)[:1000] != get(c, k).reshape(1000, -1)).any() for k in ("root_states",)):
        break
