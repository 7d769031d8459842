#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the build container only).

G1 actuator_net.npz  -- chained outputs of ATen's ``aten::lstm`` (torch.nn.LSTM, the operator the reference's
                        TorchScript ``LSTMsea`` calls) + Linear fed with the weights extracted from the
                        reference's anydrive_v3_lstm.pt; includes the SURVEY.md 8(c) seed-0 probe values.
                        The TorchScript archive itself is never loaded or executed.
G2 configs.json      -- ``class_to_dict`` of the reference's OWN config classes for the three in-scope tasks,
                        obtained by executing the reference's pure-Python config files
                        (legged_gym/envs/base/{base_config,legged_robot_config}.py, anymal_c_*_config.py,
                        cassie_config.py) under a synthetic ``legged_gym`` package (they import nothing else).
G3 models.json       -- body / DOF / shape counts, masses, foot positions derived from the reference URDFs.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get("LG_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_class_to_dict(obj):
    """helpers.py:41-56 semantics (dir() order), re-stated to avoid importing isaacgym."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [ref_class_to_dict(v) for v in val] if isinstance(val, list) else ref_class_to_dict(val)
    return out


def load_reference_configs():
    def mod(name, path=None):
        m = types.ModuleType(name)
        if path:
            m.__file__ = path
        sys.modules[name] = m
        return m

    def run(name, rel):
        path = os.path.join(REF, rel)
        m = mod(name, path)
        exec(compile(open(path).read(), path, "exec"), m.__dict__)
        return m
    for pkg in ("legged_gym", "legged_gym.envs", "legged_gym.envs.base", "legged_gym.envs.anymal_c",
                "legged_gym.envs.anymal_c.mixed_terrains", "legged_gym.envs.anymal_c.flat", "legged_gym.envs.cassie"):
        mod(pkg).__path__ = []
    bc = run("legged_gym.envs.base.base_config", "legged_gym/envs/base/base_config.py")
    sys.modules["legged_gym.envs.base"].base_config = bc
    lrc = run("legged_gym.envs.base.legged_robot_config", "legged_gym/envs/base/legged_robot_config.py")
    rough = run("legged_gym.envs.anymal_c.mixed_terrains.anymal_c_rough_config",
                "legged_gym/envs/anymal_c/mixed_terrains/anymal_c_rough_config.py")
    envs = sys.modules["legged_gym.envs"]
    envs.AnymalCRoughCfg, envs.AnymalCRoughCfgPPO = rough.AnymalCRoughCfg, rough.AnymalCRoughCfgPPO
    flat = run("legged_gym.envs.anymal_c.flat.anymal_c_flat_config", "legged_gym/envs/anymal_c/flat/anymal_c_flat_config.py")
    cas = run("legged_gym.envs.cassie.cassie_config", "legged_gym/envs/cassie/cassie_config.py")
    for pkg in ("legged_gym.envs.a1", "legged_gym.envs.anymal_b"):
        mod(pkg).__path__ = []
    a1 = run("legged_gym.envs.a1.a1_config", "legged_gym/envs/a1/a1_config.py")
    anb = run("legged_gym.envs.anymal_b.anymal_b_config", "legged_gym/envs/anymal_b/anymal_b_config.py")
    return {"anymal_c_rough": (rough.AnymalCRoughCfg, rough.AnymalCRoughCfgPPO),
            "anymal_c_flat": (flat.AnymalCFlatCfg, flat.AnymalCFlatCfgPPO),
            "cassie": (cas.CassieRoughCfg, cas.CassieRoughCfgPPO),
            "a1": (a1.A1RoughCfg, a1.A1RoughCfgPPO),
            "anymal_b": (anb.AnymalBRoughCfg, anb.AnymalBRoughCfgPPO),
            "base": (lrc.LeggedRobotCfg, lrc.LeggedRobotCfgPPO)}


def g2():
    out = {}
    for name, (E, T) in load_reference_configs().items():
        out[name] = {"env": ref_class_to_dict(E()), "train": ref_class_to_dict(T())}
    with open(os.path.join(OUT, "configs.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=False)
    print("G2 configs.json:", {k: (v["env"]["env"]["num_observations"], len([s for s in v["env"]["rewards"]["scales"].values() if s != 0])) for k, v in out.items()})


def g1():
    import torch
    from legged_games_gym_amd.utils.packing import load_actuator_weights
    w = load_actuator_weights()
    o = [0]
    def take(n, shape):
        v = torch.from_numpy(w[o[0]:o[0] + n].reshape(shape).copy()); o[0] += n; return v
    in_scale, out_scale = take(2, (2,)), take(1, (1,))
    lstm = torch.nn.LSTM(2, 8, num_layers=2, batch_first=True)
    with torch.no_grad():
        for nm, n, sh in (("weight_ih_l0", 64, (32, 2)), ("weight_hh_l0", 256, (32, 8)), ("bias_ih_l0", 32, (32,)), ("bias_hh_l0", 32, (32,)),
                          ("weight_ih_l1", 256, (32, 8)), ("weight_hh_l1", 256, (32, 8)), ("bias_ih_l1", 32, (32,)), ("bias_hh_l1", 32, (32,))):
            getattr(lstm, nm).copy_(take(n, sh))
        lw, lb = take(8, (1, 8)), take(1, (1,))
        def forward(x, hc):          # LSTMsea.forward (code/__torch__/models.py in the archive)
            y, hc = lstm(x * in_scale, hc)
            return out_scale * torch.squeeze(torch.nn.functional.linear(y, lw, lb)), hc
        torch.manual_seed(0)
        probe_x = torch.randn(24, 1, 2)
        probe_t, _ = forward(probe_x, (torch.zeros(2, 24, 8), torch.zeros(2, 24, 8)))
        # chained sequence with state carry and a mid-sequence reset of some rows (anymal.py:56-60)
        g = torch.Generator().manual_seed(123)
        R, T = 48, 8
        xs = torch.randn(T, R, 1, 2, generator=g) * torch.tensor([0.5, 6.0])
        h, c = torch.zeros(2, R, 8), torch.zeros(2, R, 8)
        taus, hs, cs = [], [], []
        for t in range(T):
            if t == 4:
                h[:, ::5] = 0.0; c[:, ::5] = 0.0
            tau, (h, c) = forward(xs[t], (h, c))
            taus.append(tau.clone()); hs.append(h.clone()); cs.append(c.clone())
    np.savez(os.path.join(OUT, "actuator_net.npz"), probe_x=probe_x.numpy(), probe_tau=probe_t.numpy(),
             survey_probe_first4=np.array([-25.7999, -8.0761, 29.0446, -5.8208], dtype=np.float32),
             xs=xs.numpy(), tau=torch.stack(taus).numpy(), h=torch.stack(hs).numpy(), c=torch.stack(cs).numpy(), reset_step=4, reset_stride=5)
    print("G1 actuator_net.npz: probe", probe_t[:4].numpy())


def g3():
    from legged_games_gym_amd.utils.model_compiler import compile_urdf
    from legged_games_gym_amd.envs import configs
    out = {}
    for stem, rel, cfg in (("anymal_c", "resources/robots/anymal_c/urdf/anymal_c.urdf", configs.AnymalCRoughCfg),
                           ("cassie", "resources/robots/cassie/urdf/cassie.urdf", configs.CassieRoughCfg),
                           ("anymal_b", "resources/robots/anymal_b/urdf/anymal_b.urdf", configs.AnymalBRoughCfg),
                           ("a1", "resources/robots/a1/urdf/a1.urdf", configs.A1RoughCfg)):
        m = compile_urdf(os.path.join(REF, rel), name=stem)
        q0 = np.array([cfg.init_state.default_joint_angles[n] for n in m.dof_names])
        feet = m.bodies_matching(cfg.asset.foot_name)
        out[stem] = {"num_bodies": m.num_bodies, "num_dof": m.num_dof, "num_shapes": m.num_shapes, "total_mass": m.total_mass,
                     "body_names": m.body_names, "dof_names": m.dof_names, "report_mass": m.report_mass.tolist(),
                     "feet_default_pose": m.report_body_positions(q0)[feet].tolist(),
                     "feet_zero_pose": m.report_body_positions(np.zeros(12))[feet].tolist(),
                     "com_default_pose": m.center_of_mass(q0).tolist()}
    with open(os.path.join(OUT, "models.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("G3 models.json:", {k: (v["num_bodies"], v["num_dof"], v["num_shapes"], round(v["total_mass"], 5)) for k, v in out.items()})


if __name__ == "__main__":
    g2(); g1(); g3()
