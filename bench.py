#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the ANYmal-C flat rollout (BASELINE.json configs[1]).

A "step" = one policy step of every env on this rank: random-init actor MLP
[48,128,64,32,12] forward + Gaussian sampling (rsl_rl ActorCritic.act) followed by
LeggedRobot.step (4 x [actuator net -> rigid-body step] + post-physics), inputs resident
in HBM.  One process per GPU; envs shard with no data-path collective in the rollout, so scaling is weak.

Launching: `python bench.py --gpus N ...` -- for N > 1 without a torch.distributed environment the parent (which never
touches a GPU) starts N worker processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, backend nccl = RCCL);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the workers are already there.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- the fused step kernel: algorithmic bytes (4022 B/env-step on flat ANYmal) x envs / mean kernel
                  duration from HIP events recorded on the launch stream around the timed graph replays themselves, against
                  the 8 TB/s HBM peak; and the second ratio SURVEY 8(d) asks for: counted fp32 FLOPs (tools/flop_count.py)
                  per second against the 157.3 TFLOP/s fp32 vector peak.
  cpu_baseline -- the CPU oracle (oracle/lg_oracle.c, OpenMP over envs, all host cores of this box) on a bounded sample of
                  the same workload; rank 0, N=1 only.
  ppo_training -- env-steps/s of the whole PPO loop (rollout + update).  At N > 1 this is the leg that carries the
                  collectives north_star names (all-gather of [returns || advantages], gradient all-reduce, mean-KL
                  all-reduce, rl/ppo.py); `rccl_ranks` is the world size as counted by a real all-reduce on the device.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, REPO)

BYTES_PER_ENV_STEP = {"anymal_c_flat": 4022, "anymal_c_rough": 4762, "cassie": 1442, "a1": 1690, "anymal_b": 4762}
HBM_PEAK_GBS = 8000.0                                                                 # MI355X_MICROARCH.md
FP32_VECTOR_PEAK_TFLOPS = 157.3                                                       # MI355X_MICROARCH.md (non-matrix fp32)
PMC_SUMMARY = os.path.join(REPO, "profiles", "r03_pmc_summary.json")     # tools/collect_r03.sh + tools/pmc_summary_r03.py: per task, the bench's own launch mode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--task", type=str, default="anymal_c_flat")
    ap.add_argument("--num-envs", type=int, default=0, help="envs per GPU (default: BASELINE.json's size for the task: 8192 for cassie, 4096 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--policy", choices=["auto", "fused", "torch"], default="auto",
                    help="auto / fused: the MFMA actor kernels (inside the step kernel for the flat actor); torch: torch ops / hipBLASLt")
    ap.add_argument("--torch-policy", action="store_true", help="same as --policy torch")
    ap.add_argument("--graph-steps", type=int, default=20, help="policy steps captured per HIP-graph replay (clamped to a divisor of --steps and --warmup)")
    ap.add_argument("--no-fused-step", action="store_true", help="keep actor kernel and step kernel separate (lg_policy_act + lg_step)")
    ap.add_argument("--f32-actor", action="store_true", help="wide actors (235/169-512-256-128): the f32-MFMA kernel instead of the split-bf16 one (lg_mlp_wide_set_precision(0))")
    ap.add_argument("--no-rollout", action="store_true", help="one launch per policy step (lg_step_policy) instead of one per --graph-steps steps (lg_rollout_policy): the A/B of the multi-step kernel")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured HIP graph per step")
    ap.add_argument("--training-iters", type=int, default=-1, help="PPO iterations timed for the ppo_training object (0 = skip; -1 = 100 for anymal_c_flat, 20 otherwise)")
    ap.add_argument("--trimesh", action="store_true", help="keep the registered mesh_type 'trimesh' (vertical faces beyond slope_treshold) instead of BASELINE.json's height-field contact")
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="a timed region shorter than this is repeated and the median reported")
    ap.add_argument("--event-steps", type=int, default=200, help="steps timed with HIP events for the step-kernel-only graph (tasks whose timed graph also holds the actor kernel)")
    a = ap.parse_args()
    if a.num_envs <= 0:
        a.num_envs = 8192 if a.task == "cassie" else 4096           # BASELINE.json configs 2/3 (4096) and 5 (8192)
    return a


PPO_LEG_FAILED = 3      # exit code of a rank whose PPO leg raised or stalled at N > 1 (after the headline line was printed)


def spawn_workers(a):
    """`python bench.py --gpus N` without a launcher: N worker processes, one per GPU, started BEFORE anything touches a GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.realpath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    if code == PPO_LEG_FAILED:   # its headline work is done; the other ranks leave the leg through their own watchdog
                        continue                 # (rank 0 still has the line to print) and the job ends non-zero
                    for q in procs:          # a dead rank leaves the others in a collective: stop them (exact PIDs)
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_workers(a))
    worker(a)


def workload_text(task, env, pol, num_envs):
    cfg = env.cfg
    # what the kernel collides against is in the bound parameters (cfg.terrain.mesh_type is restored to the registered value after make_env)
    thr = float(getattr(getattr(env, "_params", None), "hf_step_threshold", 0.0) or 0.0)
    terrain = "plane" if cfg.terrain.mesh_type in (None, "none", "plane") else \
        (f"{env.terrain.tot_rows}x{env.terrain.tot_cols} int16 curriculum height field, " +
         (f"'trimesh' contact: vertical faces beyond {thr:.3f} m per cell (slope_treshold)" if thr > 0.0 else
          "height-field contact: bilinear patches (BASELINE.json configs 3-5; the registered 'trimesh' faces with --trimesh)"))
    ctrl = "actuator-net torques" if getattr(cfg.control, "use_actuator_network", False) else f"PD control ({cfg.control.control_type})"
    sc = "self-collision ON (asset.self_collisions = 0)" if getattr(env, "self_collision_modelled", False) else \
        ("self-collision requested by the config but NOT modelled" if int(getattr(cfg.asset, "self_collisions", 1)) == 0 else "self-collision off (as configured)")
    return (f"{task}, {num_envs} envs/GPU, {terrain}, {ctrl}, {sc}, random-init policy "
            f"{[env.num_obs] + list(pol['actor_hidden_dims']) + [env.num_actions]} rollout (act = mu + sigma*eps), "
            "fixed command (0.5,0,0), obs noise + friction/mass randomisation + pushes on")


def _update_path(alg, world=1):
    tr = getattr(alg, "_mlp", None)
    if tr is None:
        return "torch autograd"
    if getattr(tr, "has_fused_minibatch", False):
        return ("lg_ppo_minibatch: forward + PPO loss + backward in one f32-MFMA kernel (v_mfma_f32_16x16x4_f32), lg_adam_step; "
                + ("one HIP graph per update" if world == 1 else "per mini-batch step: [backward HIP graph] -> one flat all-reduce (eager) -> [optimiser HIP graph]"))
    return ("wide learner kernels: chain forward k_mlp_chain_fwd64 + tiled dX / dW GEMMs (ds_read_b64_tr_b16 operand staging), split-bf16 products (hi*hi + hi*lo + lo*hi, f32 accumulate) "
            "on v_mfma_f32_32x32x16_bf16 (lg_mlp_wide_set_precision(1), the default; 0 = f32 MFMA), lg_ppo_loss, lg_adam_step")


def worker(a):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = None
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
    coll_dev = dev if (world == 1 or backend == "nccl") else torch.device("cpu")
    ranks_seen = 1
    if world > 1:                       # did the collective library see N ranks?  one real all-reduce answers it
        ones = torch.ones(1, device=coll_dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    # torch's CUDA generator creates its graph-safe state at the first capture in the process; done under inference_mode (the
    # rollout graphs below) those tensors could not be touched by the later training leg.  Prime them in normal mode, keep alive.
    rng_prime = torch.cuda.CUDAGraph()
    with torch.cuda.graph(rng_prime, capture_error_mode="thread_local"):
        torch.zeros(1, device=dev).add_(1.0)

    import contextlib
    import io
    from legged_games_gym_amd.envs import task_registry  # registers tasks
    from legged_games_gym_amd.utils import get_args
    from legged_games_gym_amd.rl import ActorCritic
    from legged_games_gym_amd.utils.helpers import class_to_dict

    if a.f32_actor:
        from legged_games_gym_amd import capi as _capi
        _capi.load_library().lg_mlp_wide_set_precision(0)
    args = get_args(["--task", a.task, "--num_envs", str(a.num_envs), "--headless", "--sim_device", f"cuda:{local_rank}",
                     "--rl_device", f"cuda:{local_rank}"])
    env_cfg, train_cfg = task_registry.get_cfgs(a.task)
    env_cfg.seed = train_cfg.seed + rank                  # rank-local RNG stream
    mesh_registered = env_cfg.terrain.mesh_type
    if mesh_registered == "trimesh" and not a.trimesh:    # BASELINE.json configs 3-5 name HEIGHT-FIELD contact
        env_cfg.terrain.mesh_type = "heightfield"
    with contextlib.redirect_stdout(io.StringIO()):
        env, _ = task_registry.make_env(a.task, args, env_cfg=env_cfg)
    env_cfg.terrain.mesh_type = mesh_registered           # (the registry hands out its one cfg object: leave it as registered)
    env.set_fixed_commands(0.5, 0.0, 0.0)                 # "fixed command" of BASELINE.json
    torch.manual_seed(train_cfg.seed)                     # random-init policy, same on every rank
    pol = class_to_dict(train_cfg.policy)
    policy = ActorCritic(env.num_obs, env.num_obs, env.num_actions, **pol).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        obs, _ = env.reset()
    use_torch = a.torch_policy or a.policy == "torch"
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
    rollout_kernel = False                                # ... and G steps per launch (lg_rollout_policy)
    # policy steps per graph replay / per rollout launch: a divisor of the TIMED step count only; the part of the warm-up that is not a
    # multiple of it runs as eager single steps first (the driver's `--steps 20 --warmup 5` then still times ONE 20-step launch per region)
    G = max(g for g in range(1, max(1, a.graph_steps) + 1) if a.steps % g == 0)
    with torch.inference_mode():
        for _ in range(a.warmup % G):
            env.step(policy_act(env.obs_buf))
        if not use_torch and not a.no_fused_step:
            try:
                if not a.no_rollout and G > 1:
                    try:                                       # G policy steps per LAUNCH (lg_rollout_policy), one HIP graph per segment
                        if a.no_graph:                         # (eager: every dispatch gets its own counter row under rocprofv3 --pmc)
                            _roll_storage = env.rollout_policy(fused, G)
                            one_step = lambda: env.rollout_policy(fused, G, storage=_roll_storage)
                        else:
                            one_step, _roll_storage = env.make_graphed_rollout(fused, G)
                        rollout_kernel = True
                    except RuntimeError as exc:
                        if "multi-step rollout kernel" not in str(exc):
                            raise
                if not rollout_kernel:
                    if a.no_graph:
                        G = 1
                    one_step = env.make_graphed_policy_step(fused, steps_per_replay=G) if not a.no_graph else (lambda: env.step_policy(fused))
                    if a.no_graph:
                        one_step()
                fused_step = True
            except RuntimeError as exc:
                if "fused policy step" not in str(exc):    # only "no fused kernel for this sim / actor pair" is a fall-back case
                    raise
                fused_step = False
        if fused_step:
            pass
        elif a.no_graph:
            G = 1

            def one_step():
                env.step(policy_act(env.obs_buf))             # the full VecEnv step (one lg_step call)
        else:
            one_step = env.make_graphed_step(policy_act, steps_per_replay=G)      # policy forward + sampling + lg_step in ONE HIP graph
        for _ in range(a.warmup // G):
            one_step()

        def timed_region():
            """EXACTLY a.steps policy steps between barrier + synchronize; HIP events on the launch stream bracket the same replays."""
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(a.steps // G):                  # G policy steps per graph replay
                one_step()
            e1.record()
            sync()
            return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3

        walls, evs = [], []
        w, e = timed_region()
        walls.append(w); evs.append(e)
        # short run (the driver's --steps 20): repeat the region, report the median.  Rank 0 decides (regions contain barriers).
        repeats = 1 if w * 1e3 >= a.min_timed_ms else int(min(400, max(9, round(0.5 / max(w, 1e-6)))))
        if world > 1:
            flag = torch.tensor([repeats], device=coll_dev)
            dist.broadcast(flag, src=0)
            repeats = int(flag.item())
        for _ in range(repeats - 1):
            w, e = timed_region()
            walls.append(w); evs.append(e)

        # step kernel alone, for tasks whose timed graph also holds the actor kernel: HIP events around replays of a graph that holds
        # ONLY k_step, KG launches back to back, directly after the timed region
        if fused_step and not a.no_graph:
            kern_ms_each = [1e3 * e / a.steps for e in evs]     # the timed graph IS G x k_step<..., POL>: same replays, same region
            kern_method = (f"HIP events on the launch stream around the {a.steps // G} graph replays of the timed region itself "
                           f"(median of {repeats} regions; includes the ~1 us gaps between launches" + ")")
        else:
            KG = 20
            fixed_actions = (policy_act(env.obs_buf) if not fused_step else fused.act(env.obs_buf)).clone()
            kernel_replay = env.make_graphed_step(lambda _obs: fixed_actions, steps_per_replay=KG)   # KG x lg_step, nothing else
            n_rep = max(3, a.event_steps // KG)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_rep)]
            kernel_replay()
            for s_, e_ in ev:
                s_.record()
                kernel_replay()
                e_.record()
            torch.cuda.synchronize()
            kern_ms_each = [s_.elapsed_time(e_) / KG for s_, e_ in ev]
            kern_method = f"HIP events around {n_rep} replays of a HIP graph of {KG} back-to-back launches of the step kernel alone, right after the timed region (median per-launch time)"
    finite = bool(torch.isfinite(env.obs_buf).all()) and bool(torch.isfinite(env.root_states).all())
    kern_ms = statistics.median(kern_ms_each)
    t = torch.tensor(walls, dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # per region: the slowest rank
    elapsed = float(t.median().item()) if repeats > 1 else float(t[0].item())

    # cost of self-collision (on in the anymal_c_flat config): the same rollout with asset.self_collisions switched off
    sc_cost = None
    if getattr(env, "self_collision_modelled", False) and world == 1 and fused_step and not a.no_graph:
        try:
            env_cfg2, _ = task_registry.get_cfgs(a.task)
            keep = env_cfg2.asset.self_collisions
            env_cfg2.asset.self_collisions = 1
            with contextlib.redirect_stdout(io.StringIO()):
                env2, _ = task_registry.make_env(a.task, args, env_cfg=env_cfg2)
                env2.set_fixed_commands(0.5, 0.0, 0.0)
                env2.reset()
            env_cfg2.asset.self_collisions = keep
            with torch.inference_mode():
                fused2 = FusedActor(policy, dev, seed=train_cfg.seed + rank, step_counter=env2._sim.buf["step_counter"])
                step2 = env2.make_graphed_policy_step(fused2, steps_per_replay=20)
                for _ in range(5):
                    step2()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    step2()
                torch.cuda.synchronize()
                ms_off = 1e3 * (time.perf_counter() - t0) / 1000
            sc_cost = {"ms_per_step_without_self_collision": ms_off, "note": "same rollout, asset.self_collisions = 1 (1000 steps)"}
            del env2, fused2, step2
        except Exception as exc:
            sc_cost = {"error": f"{type(exc).__name__}: {exc}"}

    out = None
    if rank == 0:
        from tools.flop_count import flops_per_env_step
        total_envs = a.num_envs * world
        value = total_envs * a.steps / elapsed
        bpe = BYTES_PER_ENV_STEP.get(a.task, 4022)
        traffic, mfma_busy, pm_src = None, None, None   # PMC figures per k_step launch from the committed rocprofv3 passes (same workload only)
        try:
            pm = json.load(open(PMC_SUMMARY))["tasks"].get(a.task)
            if pm and pm["envs_per_gpu"] == a.num_envs:       # counters of the same workload only; per launch of the dominant kernel, like `achieved`
                per_step = pm["k_step"]["steps_per_launch"]
                traffic = pm["k_step"]["traffic_bytes_per_launch"] / per_step * (G if rollout_kernel else 1)
                mfma_busy = 100.0 * pm["k_step"]["mfma_busy_frac_of_busy_cycles"]
                pm_src = (os.path.relpath(PMC_SUMMARY, REPO) + f" (rocprofv3 --pmc passes of `{pm['command']}`: FETCH_SIZE / WRITE_SIZE KiB per dispatch of {pm['k_step']['kernel']}, "
                          f"2*FETCH+WRITE, {per_step} policy step(s) per dispatch; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES)")
        except Exception:
            pass
        achieved = bpe * a.num_envs / (kern_ms * 1e-3) / 1e9
        fl = flops_per_env_step(a.task, int(env.cfg.control.decimation))
        step_flops = fl["total"] - (0 if fused_step else fl["actor_mlp"])          # what the timed kernel computes
        tflops = step_flops * a.num_envs / (kern_ms * 1e-3) / 1e12
        kname = {"anymal_c_flat": "k_step<AnymalTraits,NET,plane" + ((",POL,ROLL> (actor + step, %d steps per launch)" % G) if rollout_kernel else ",POL> (actor + step)" if fused_step else ">"),
                 "cassie": "k_step<CassieTraits,PD,HF>"}.get(a.task, "k_step<AnymalTraits," + ("NET" if getattr(env.cfg.control, "use_actuator_network", False) else "PD") + ",HF>")
        wide_bf16 = (not use_torch) and (not fused_step) and capi_wide_precision() == 1
        dtype_text = "f32" if not wide_bf16 else "f32 physics + bf16x3 actor (split-bf16 products hi*hi + hi*lo + lo*hi with f32 accumulation: 16 significand bits per operand; --f32-actor times the f32-MFMA actor)"
        out = {
            "metric": "env-steps/sec (whole node), ANYmal-C flat 4096 envs/GPU" if a.task == "anymal_c_flat" else f"env-steps/sec (whole node), {a.task}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_text, "data": "synthetic",
            "config": {"workload": workload_text(a.task, env, pol, a.num_envs),
                       "envs_per_gpu": a.num_envs, "decimation": int(env.cfg.control.decimation), "sim_dt": float(env.sim_params.dt),
                       "parallelism": f"env-sharded x{world}", "state_finite": finite,
                       "launch": (f"lg_rollout_policy: ONE launch of k_step<...,POL,ROLL> per {G} policy steps (actor fused into the step, every workgroup walks through the {G} steps of its own 16 envs; "
                                  f"per-step obs / actions / rewards / dones to rollout storage), " + ("eager launches" if a.no_graph else "replayed as a HIP graph") + "; the last workgroup to finish publishes extras[\"episode\"]") if rollout_kernel else
                                 (("eager" if a.no_graph else f"HIP graph of {G} policy steps per replay") + (": ONE kernel per step, actor fused into the step (lg_step_policy)" if fused_step else " (policy + lg_step)")
                                  + ("" if a.no_graph or os.environ.get("LG_DEFER_EXTRAS", "1") == "0" else "; extras[\"episode\"] deferred to the next launch, one lg_extras_flush node per replay")),
                       "policy": "torch ops (hipBLASLt)" if use_torch else ("MFMA actor inside k_step (v_mfma_f32_16x16x4_f32, 4 waves)" if fused_step else "actor kernel lg_policy_act: k_policy_act_wide, 32 envs per workgroup, split-bf16 products (hi*hi + hi*lo + lo*hi, f32 accumulate) on v_mfma_f32_32x32x16_bf16")},
            "repeats": repeats,
            "timing": f"median of {repeats} timed regions of exactly {a.steps} steps each (a region shorter than {a.min_timed_ms:g} ms is repeated)" if repeats > 1 else f"one timed region of {a.steps} steps",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": pm_src, "kernel": kname,
                         "policy_steps_per_launch": G if rollout_kernel else 1,
                         "algorithmic_bytes_per_launch": bpe * a.num_envs * (G if rollout_kernel else 1), "launch_ms": kern_ms * (G if rollout_kernel else 1),
                         "kernel_ms": kern_ms, "kernel_ms_method": kern_method,
                         "algorithmic_bytes_per_env_step": bpe,
                         "flops": {"per_env_step": step_flops, "breakdown": fl, "achieved_tflops": tflops, "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                                   "frac": tflops / FP32_VECTOR_PEAK_TFLOPS, "mfma_busy_pct": mfma_busy,
                                   "note": "counted fp32 flops of the timed kernel (tools/flop_count.py: ABA + contacts, actuator LSTM, actor when fused) / fp32 vector peak"},
                         "note": "issue/latency-bound, not HBM-bound: one 512-register rigid-body wave per SIMD runs a serial chain of ~20k instructions; state is L2/MALL-resident; see DESIGN.md section 5"},
        }
        if sc_cost is not None:
            if "ms_per_step_without_self_collision" in sc_cost:
                sc_cost["self_collision_cost_ms_per_step"] = out["ms_per_step"] - sc_cost["ms_per_step_without_self_collision"]
            out["self_collision"] = sc_cost
        if world > 1:
            out["rccl_ranks" if backend == "nccl" else f"{backend}_ranks"] = ranks_seen

    # The PPO leg is extra information and is never allowed to take the headline line down.  At N > 1 it contains collectives, so
    # a rank that fails or stalls inside it would leave the others waiting: every rank runs it under a watchdog, and after a
    # failure no further collective is attempted -- rank 0 prints the line it already holds (with the error) and the ranks leave.
    printed = threading.Lock()

    def emit(training):
        if rank != 0 or not printed.acquire(blocking=False):
            return
        if training is not None:
            out["ppo_training"] = training
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, env)
        sys.__stdout__.write(json.dumps(out) + "\n")         # (the watchdog may fire while the leg has sys.stdout redirected)
        sys.__stdout__.flush()

    training, broken = None, False
    iters = a.training_iters if a.training_iters >= 0 else (100 if a.task == "anymal_c_flat" else 20)
    if iters > 0:
        limit = float(os.environ.get("LG_BENCH_PPO_TIMEOUT_S", "300"))
        finished = threading.Event()

        def abandon():
            if finished.is_set():
                return
            emit({"error": f"PPO leg did not finish within {limit:g} s at world size {world}: abandoned, headline line unaffected"})
            sys.stdout.flush()
            os._exit(PPO_LEG_FAILED)                  # the line is out, but a stalled collective is not a success: non-zero for the driver
        timer = threading.Timer(limit, abandon)
        timer.daemon = True
        if world > 1:
            timer.start()
        try:
            training = training_leg(a, iters, rank, local_rank, world, coll_dev, ranks_seen, backend)
        except Exception as exc:
            training, broken = {"error": f"{type(exc).__name__}: {exc}"}, world > 1
        finally:
            finished.set()
            timer.cancel()
    emit(training)
    if world > 1:
        if broken:                               # the other ranks may still sit in a collective of the leg: do not join them again
            sys.stdout.flush()
            os._exit(PPO_LEG_FAILED)
        dist.barrier()
        dist.destroy_process_group()


def training_leg(a, iters, rank, local_rank, world, coll_dev, ranks_seen, backend):
    """Not the headline metric: env-steps/s of the whole PPO loop (24-step rollouts + 5 x 4 mini-batch updates per iteration,
    reference train cfg) with the bundled runner -- rollout graph + lg_rollout_record, update on the MFMA learner kernels where
    the networks have a compiled shape.  At world > 1 every iteration contains the RCCL all-gather of [returns || advantages]
    and, per mini-batch, the gradient and mean-KL all-reduces (rl/ppo.py); time = max over ranks between barriers."""
    import contextlib
    import io
    import torch
    import torch.distributed as dist
    from legged_games_gym_amd.envs import task_registry
    from legged_games_gym_amd.utils import get_args

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    fault = os.environ.get("LG_BENCH_PPO_FAULT", "")             # test hook (tests/test_gpu_bench.py): "raise:<rank>" / "hang:<rank>"
    if fault and int(fault.split(":")[1]) == rank:
        if fault.startswith("raise"):
            raise RuntimeError("injected PPO-leg failure")
        time.sleep(3600)
    with contextlib.redirect_stdout(io.StringIO()):              # rank 0 prints ONE line: everything here stays silent
        args = get_args(["--task", a.task, "--headless", "--sim_device", f"cuda:{local_rank}", "--rl_device", f"cuda:{local_rank}", "--num_envs", str(a.num_envs)])
        env_cfg, train_cfg = task_registry.get_cfgs(a.task)
        env_cfg.seed = train_cfg.seed + rank
        env, _ = task_registry.make_env(a.task, args, env_cfg=env_cfg)
        runner, train_cfg = task_registry.make_alg_runner(env, a.task, args, log_root=None)
        runner.learn(num_learning_iterations=6, init_at_random_ep_len=True)      # eager warm-up update, graph captures
        sync()
        t0 = time.perf_counter()
        runner.learn(num_learning_iterations=iters)
        sync()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    T = int(train_cfg.runner.num_steps_per_env)
    alg = runner.alg
    out = {"value": world * a.num_envs * T * iters / dt, "unit": "env-steps/s (rollout + PPO update, whole job)", "iterations": iters,
           "ms_per_iteration": 1e3 * dt / iters, "steps_per_env": T, "epochs_x_minibatches": [alg.num_learning_epochs, alg.num_mini_batches],
           "update_path": _update_path(alg, world),
           "final_learning_rate": float(alg.learning_rate),
           "note": "includes re-capturing the rollout graph at the start of the timed learn() call"}
    if world > 1:
        steps = alg.num_learning_epochs * alg.num_mini_batches
        if getattr(alg, "_gflat", None) is not None:      # kernel path: every .grad is a view of one buffer whose last slot carries the KL
            coll = {"all_gather_returns_advantages": 1, "flat_all_reduce_all_gradients_and_kl": steps}
        else:
            coll = {"all_gather_returns_advantages": 1, "gradient_all_reduce": steps, "kl_all_reduce": steps if alg.schedule == "adaptive" else 0}
        coll.update({"backend": "nccl (RCCL)" if backend == "nccl" else backend, "ranks": ranks_seen})
        out["collectives_per_iteration"] = coll
        out["host_launches_per_update"] = getattr(alg, "dp_launches", None) or {"note": "eager kernel launches (two-graph capture not used: " + str(getattr(alg, "_dp_graph_error", "torch MLP path")) + ")"}
    return out


def capi_wide_precision():
    """Current lg_mlp_wide_set_precision mode (0 = f32 MFMA, 1 = split-bf16) without changing it."""
    from legged_games_gym_amd import capi
    lib = capi.load_library()
    mode = lib.lg_mlp_wide_set_precision(-1)          # an invalid mode only returns the current one
    return int(mode)


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


def cpu_baseline(a, env):
    """Time the CPU oracle on the same workload (own restatement: PhysX CPU is unavailable): the GPU leg's terrain (the same int16 height
    field, the same contact rule: bilinear patches, or vertical faces with --trimesh), env origins / terrain levels, friction and mass."""
    import numpy as np
    from tests.common import make_setup, grid_origins
    from oracle.oracle import OracleSim
    cores = _usable_cores()
    N = a.num_envs
    terr = getattr(env, "terrain", None) if env.cfg.terrain.mesh_type in ("heightfield", "trimesh") else None
    mesh = "trimesh" if a.trimesh else "heightfield"

    def tweak(cfg):
        if terr is not None:
            cfg.terrain.mesh_type = mesh if cfg.terrain.mesh_type == "trimesh" else cfg.terrain.mesh_type
    cfg, robot, p, names, model, w = make_setup(a.task, N, plane=(terr is None), terrain=terr, tweak=tweak)
    o = OracleSim(p, model, robot, w, threads=cores)
    if terr is not None:
        o.set_terrain(terr.heightsamples, terr.env_origins)
        for k in ("terrain_levels", "terrain_types"):
            o.buf[k][:] = env._sim.buf[k].cpu().numpy()
    o.buf["env_origins"][:] = env._sim.buf["env_origins"].cpu().numpy()
    o.buf["friction_coeffs"][:] = env._sim.buf["friction_coeffs"].cpu().numpy()
    o.buf["base_mass_delta"][:] = env._sim.buf["base_mass_delta"].cpu().numpy()
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
            "sample": f"{N} envs x {steps} policy steps of the same workload (env step only, N(0,1) actions) on "
                      + ("the plane" if terr is None else f"the GPU leg's {terr.tot_rows}x{terr.tot_cols} int16 curriculum height field ({'vertical-face' if p.hf_step_threshold > 0 else 'bilinear height-field'} contact, same env origins / terrain levels)")
                      + ", the GPU leg's friction / base-mass randomisation; oracle/lg_oracle.c with OpenMP over envs; PhysX CPU path of the reference is not runnable here"}


if __name__ == "__main__":
    main()
