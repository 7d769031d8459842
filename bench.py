#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the ANYmal-C flat rollout (BASELINE.json configs[1]).

A "step" = one policy step of every env on this rank: random-init actor MLP
[48,128,64,32,12] forward + Gaussian sampling (rsl_rl ActorCritic.act) followed by
LeggedRobot.step (4 x [actuator net -> rigid-body step] + post-physics), inputs resident
in HBM.  One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the env); envs shard with no
data-path collective, so scaling is weak; the only collectives are the timing barrier / max.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the fused step kernel: algorithmic bytes (4022 B/env-step on
                  flat ANYmal) x envs / mean kernel duration from HIP events recorded on the
                  launch stream inside the timed region, against the 8 TB/s HBM peak.
  cpu_baseline -- the CPU oracle (oracle/lg_oracle.c, OpenMP over envs, all host cores of
                  this box) on a bounded sample of the same workload; rank 0, N=1 only.
and, as extra information (N=1, anymal_c_flat; not the headline metric):
  ppo_training -- env-steps/s of the whole PPO loop (rollout graph + kernel update), --training-iters iterations.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, REPO)

BYTES_PER_ENV_STEP = {"anymal_c_flat": 4022, "anymal_c_rough": 4762, "cassie": 1442}
HBM_PEAK_GBS = 8000.0                                                                 # MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--task", type=str, default="anymal_c_flat")
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--policy", choices=["auto", "fused", "torch"], default="auto",
                    help="auto: fused MFMA actor kernel for narrow nets (hidden <= 128, the flat config), torch/hipBLASLt otherwise")
    ap.add_argument("--torch-policy", action="store_true", help="same as --policy torch")
    ap.add_argument("--graph-steps", type=int, default=20, help="policy steps captured per HIP-graph replay (clamped to a divisor of --steps and --warmup)")
    ap.add_argument("--no-fused-step", action="store_true", help="keep actor kernel and step kernel separate (lg_policy_act + lg_step)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured HIP graph per step")
    ap.add_argument("--training-iters", type=int, default=100, help="PPO iterations timed for the extra ppo_training object (0 = skip; N=1, anymal_c_flat only)")
    ap.add_argument("--event-steps", type=int, default=200, help="eager steps timed with HIP events for the roofline object")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("LG_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on a 1-GPU box
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    # torch's CUDA generator creates its graph-safe state at the first capture in the process; done under inference_mode (the
    # rollout graphs below) those tensors could not be touched by the later training leg.  Prime them in normal mode, keep alive.
    rng_prime = torch.cuda.CUDAGraph()
    with torch.cuda.graph(rng_prime):
        torch.zeros(1, device=dev).add_(1.0)

    import contextlib
    import io
    from legged_games_gym_amd.envs import task_registry  # registers tasks
    from legged_games_gym_amd.utils import get_args
    from legged_games_gym_amd.rl import ActorCritic
    from legged_games_gym_amd.utils.helpers import class_to_dict

    args = get_args(["--task", a.task, "--num_envs", str(a.num_envs), "--headless", "--sim_device", f"cuda:{local_rank}",
                     "--rl_device", f"cuda:{local_rank}"])
    env_cfg, train_cfg = task_registry.get_cfgs(a.task)
    env_cfg.seed = train_cfg.seed + rank                  # rank-local RNG stream
    with contextlib.redirect_stdout(io.StringIO()):
        env, _ = task_registry.make_env(a.task, args, env_cfg=env_cfg)
    env.set_fixed_commands(0.5, 0.0, 0.0)                 # "fixed command" of BASELINE.json
    torch.manual_seed(train_cfg.seed)                     # random-init policy, same on every rank
    pol = class_to_dict(train_cfg.policy)
    policy = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **pol).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        obs, _ = env.reset()
    use_torch = a.torch_policy or a.policy == "torch"      # "auto" = the MFMA actor (all compiled shapes beat torch/hipBLASLt since the weight prefetch)
    a.torch_policy = use_torch
    if use_torch:
        policy_act = policy.act
    else:
        from legged_games_gym_amd.rl import FusedActor
        fused = FusedActor(policy, dev, seed=train_cfg.seed + rank, step_counter=env._sim.buf["step_counter"])
        policy_act = fused.act

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fused_step = False                                    # actor fused INTO the step kernel (lg_step_policy): flat actor only
    G = 1
    if not a.no_graph:
        G = max(g for g in range(1, max(1, a.graph_steps) + 1) if a.steps % g == 0 and a.warmup % g == 0)
    with torch.inference_mode():
        if not use_torch and not a.no_fused_step:
            try:
                one_step = env.make_graphed_policy_step(fused, steps_per_replay=G) if not a.no_graph else (lambda: env.step_policy(fused))
                if a.no_graph:
                    one_step()
                fused_step = True
            except RuntimeError:
                fused_step = False
        if fused_step:
            pass
        elif a.no_graph:
            def one_step():
                env.step(policy_act(env.obs_buf))             # the full VecEnv step (one lg_step call)
        else:
            one_step = env.make_graphed_step(policy_act, steps_per_replay=G)      # policy forward + sampling + lg_step in ONE HIP graph
        for _ in range(a.warmup // G):
            one_step()
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps // G):                      # exactly a.steps policy steps: G per graph replay
            one_step()
        sync()
        elapsed = time.perf_counter() - t0
        # roofline: duration of the step kernel from HIP events (recorded on the launch stream) around replays of a HIP graph
        # that holds ONLY that kernel, G launches back to back, directly after the timed region.  (Events around eager
        # launches measure the host's enqueue latency instead: the kernel is shorter than one Python call.)
        if fused_step and not a.no_graph:
            kernel_replay, KG = one_step, G                      # the timed graph already is G x k_step<..., POL>
        else:
            KG = 20
            fixed_actions = (policy_act(env.obs_buf) if not fused_step else fused.act(env.obs_buf)).clone()
            kernel_replay = env.make_graphed_step(lambda _obs: fixed_actions, steps_per_replay=KG)   # G x lg_step, nothing else
        n_rep = max(1, min(a.event_steps, a.steps) // KG)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_rep)]
        kernel_replay()
        for s_, e_ in ev:
            s_.record()
            kernel_replay()
            e_.record()
        torch.cuda.synchronize()
    finite = bool(torch.isfinite(env.obs_buf).all()) and bool(torch.isfinite(env.root_states).all())
    kern_ms = sum(s_.elapsed_time(e_) for s_, e_ in ev) / (n_rep * KG)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    out = None
    if rank == 0:
        total_envs = a.num_envs * world
        value = total_envs * a.steps / elapsed
        bpe = BYTES_PER_ENV_STEP.get(a.task, 4022)
        traffic = None                      # PMC bytes per k_step launch from the committed rocprofv3 passes (same workload only)
        try:
            pm = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_summary.json")))
            if a.task == "anymal_c_flat" and a.num_envs == 4096:
                traffic = pm["k_step_traffic_bytes"]["fetch_doubled_sum"]
        except Exception:
            pass
        achieved = bpe * a.num_envs / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (whole node), ANYmal-C flat 4096 envs/GPU" if a.task == "anymal_c_flat" else f"env-steps/sec (whole node), {a.task}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.task}, {a.num_envs} envs/GPU, plane, actuator-net torques, random-init policy "
                                   f"{[env.num_obs] + list(pol['actor_hidden_dims']) + [env.num_actions]} rollout (act = mu + sigma*eps), "
                                   "fixed command (0.5,0,0), obs noise + friction/mass randomisation + pushes on",
                       "envs_per_gpu": a.num_envs, "decimation": int(env.cfg.control.decimation), "sim_dt": float(env.sim_params.dt),
                       "parallelism": f"env-sharded x{world}", "state_finite": finite,
                       "launch": ("eager" if a.no_graph else f"HIP graph of {G} policy steps per replay") + (": ONE kernel, actor fused into the step (lg_step_policy)" if fused_step else " (policy + lg_step)"),
                       "policy": "torch ops (hipBLASLt)" if a.torch_policy else ("MFMA actor inside k_step (v_mfma_f32_16x16x4_f32, 4 waves)" if fused_step else "fused MFMA actor kernel (lg_policy_act, v_mfma_f32_16x16x4_f32)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE KiB)" if traffic else None, "kernel": ("k_step<AnymalTraits,NET,plane,POL> (actor + step)" if fused_step else "k_step<AnymalTraits,NET,plane>") if a.task != "cassie" else "k_step<CassieTraits>",
                         "kernel_ms": kern_ms, "kernel_ms_method": f"HIP events around {n_rep} replays of a HIP graph of {KG} back-to-back launches of the step kernel alone, right after the timed region (per-launch average, includes ~1 us launch gap)",
                         "algorithmic_bytes_per_env_step": bpe,
                         "note": "issue/latency-bound, not HBM-bound: 4096 envs = 256 workgroups x (1 rigid-body + 3 helper waves), one wave per SIMD, ~20k serial instructions on the rigid-body wave at one per ~6.5 cycles; see DESIGN.md section 5"},
        }
        if world == 1 and a.training_iters > 0 and a.task == "anymal_c_flat":
            try:                                 # extra information, never allowed to take the headline line down
                out["ppo_training"] = training_leg(a)
            except Exception as exc:
                out["ppo_training"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def training_leg(a):
    """Not the headline metric: env-steps/s of the whole PPO loop (24-step rollouts + 5 x 4 mini-batch updates per iteration,
    reference anymal_c_flat train cfg) with the bundled runner -- rollout graph of lg_step_policy + lg_rollout_record, update =
    lg_mlp_forward / lg_ppo_loss / lg_mlp_backward / lg_adam_step replayed as one HIP graph per mini-batch."""
    import torch
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):              # rank 0 prints ONE line: everything here stays silent
        args = get_args(["--task", a.task, "--headless", "--sim_device", "cuda:0", "--rl_device", "cuda:0", "--num_envs", str(a.num_envs)])
        env, _ = task_registry.make_env(a.task, args)
        runner, train_cfg = task_registry.make_alg_runner(env, a.task, args, log_root=None)
        runner.learn(num_learning_iterations=6, init_at_random_ep_len=True)      # eager warm-up update, graph captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.learn(num_learning_iterations=a.training_iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    T = int(train_cfg.runner.num_steps_per_env)
    alg = runner.alg
    return {"value": a.num_envs * T * a.training_iters / dt, "unit": "env-steps/s (rollout + PPO update)", "iterations": a.training_iters,
            "ms_per_iteration": 1e3 * dt / a.training_iters, "steps_per_env": T, "epochs_x_minibatches": [alg.num_learning_epochs, alg.num_mini_batches],
            "update_path": "MFMA learner kernels" if getattr(alg, "_mlp", None) is not None else "torch autograd",
            "final_learning_rate": float(alg.learning_rate),
            "note": "includes re-capturing the rollout graph at the start of the timed learn() call"}


def _usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LG_BENCH_MAX_THREADS", "64"))))


def cpu_baseline(a):
    """Time the CPU oracle on the same workload shape (own restatement: PhysX CPU is unavailable)."""
    import numpy as np
    from tests.common import make_setup, grid_origins, randomize_env_params
    from oracle.oracle import OracleSim
    cores = _usable_cores()
    N = a.num_envs
    cfg, robot, p, names, model, w = make_setup(a.task, N)
    o = OracleSim(p, model, robot, w, threads=cores)
    o.buf["env_origins"][:] = grid_origins(N)
    fr, dm = randomize_env_params(N, 1)
    o.buf["friction_coeffs"][:] = fr
    o.buf["base_mass_delta"][:] = dm
    o.reset_idx(np.arange(N, dtype=np.int32), 0)
    rng = np.random.default_rng(0)
    acts = rng.standard_normal((8, N, 12)).astype(np.float32)
    t0 = time.perf_counter()
    o.step(acts[0], 1); o.step(acts[1], 2)
    per = (time.perf_counter() - t0) / 2
    steps = int(max(3, min(2000, a.cpu_seconds / max(per, 1e-6))))
    t0 = time.perf_counter()
    for i in range(steps):
        o.step(acts[i % 8], 3 + i)
    dt = time.perf_counter() - t0
    return {"value": N * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{N} envs x {steps} policy steps of the same workload (env step only, N(0,1) actions), "
                      f"oracle/lg_oracle.c with OpenMP over envs; PhysX CPU path of the reference is not runnable here"}


if __name__ == "__main__":
    main()
