#!/usr/bin/env python3
"""Algorithmic fp32 FLOPs of one env-step of the fused kernel ("physics: builder counts its own ABA +
contact FLOPs"), used by bench.py for the second roofline ratio (achieved FLOP/s / fp32 vector peak).

The count follows the structure of the rigid-body step as both oracle/lg_oracle.c:physics_substep_env and
csrc/lg_kernels.hip:physics_substep implement it, with one flop per add / mul (an FMA counts 2), per ENV (work every
lane of an env repeats -- the 6 x 6 base solve -- is counted once).  Primitive costs:

    dot 5, cross 9, scale / add 3, 3x3 * vec 15, 3x3 * 3x3 45,
    ai_add 21, ai_add_point 38, ai_add_rank1 63, ai_mul (6x6 * 6) 66, ai_mul_w (6x6 * (w,0)) 30, ai_shift 114,
    solve6 (LDL^T + two triangular solves) 230, body_terms 233, fk_joint + velocity-product terms 209.

Data-dependent work (contacts that are switched on) is counted for `active_contacts` points per env -- 4 by default: a
standing quadruped's feet (2 for the biped).
"""

PRIM = dict(dot=5, cross=9, vec=3, mv=15, mm=45, ai_add=21, ai_add_point=38, ai_add_rank1=63, ai_mul=66, ai_mul_w=30,
            ai_shift=114, solve6=230, body_terms=233, fk_joint=209, quat_to_mat=27)

ROBOTS = {      # K limbs x L joints, collision points per limb / on the base
    "anymal": dict(K=4, L=3, limb_points=5, base_points=2, active=4),
    "cassie": dict(K=2, L=6, limb_points=2, base_points=1, active=2),
}


def physics_flops_per_substep(robot="anymal", heightfield=False, passes=2, active_contacts=None):
    r, P = ROBOTS[robot], PRIM
    nj = r["K"] * r["L"]
    npts = r["K"] * r["limb_points"] + r["base_points"]
    act = r["active"] if active_contacts is None else active_contacts
    kin = P["quat_to_mat"] + nj * P["fk_joint"]
    bodies = (nj + 1) * P["body_terms"]
    pts = npts * (P["mv"] + 2 * P["vec"] + P["cross"] + 30 + (30 if heightfield else 0))     # position, velocity, contact_setup (+ bilinear patch)
    joint_in = P["ai_add"] + 6 + P["ai_mul_w"] + 30 + P["ai_add_rank1"] + P["ai_mul"] + 19 + P["ai_shift"]      # inward recursion per joint
    joint_out = 18 + 13 + 6 + 8                                                                                # outward recursion per joint
    base = 27 * (r["K"] - 1) + P["ai_add"] + 6 + P["solve6"]
    contact = (20 + P["ai_add_point"] + 2 * P["cross"] + P["ai_add_rank1"] + 6) + 60                            # assemble + evaluate, when on
    per_pass = nj * (joint_in + joint_out) + base + act * contact
    integrate = 4 * nj + 60
    return kin + bodies + pts + passes * per_pass + integrate


def flops_per_env_step(task="anymal_c_flat", decimation=4):
    """{component: flops} per env-step for a BASELINE.json task name."""
    robot = "cassie" if task == "cassie" else "anymal"
    hf = task != "anymal_c_flat"
    out = {"physics": decimation * physics_flops_per_substep(robot, hf), "post_physics": 2000}
    if task.startswith("anymal"):
        out["actuator_lstm"] = 840 * 2 * 12 * decimation       # 840 MAC x 12 joints x 4 sub-steps = 80 640 flop
    dims = {"anymal_c_flat": [48, 128, 64, 32, 12], "anymal_c_rough": [235, 512, 256, 128, 12], "cassie": [169, 512, 256, 128, 12],
            "a1": [235, 512, 256, 128, 12], "anymal_b": [235, 512, 256, 128, 12]}[task]
    out["actor_mlp"] = 2 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    if hf:
        out["height_sampling"] = (187 if robot == "anymal" else 121) * 25
    out["total"] = sum(out.values())
    return out


if __name__ == "__main__":
    import json
    print(json.dumps({t: flops_per_env_step(t) for t in ("anymal_c_flat", "anymal_c_rough", "cassie")}, indent=1))
