"""-m gpu: bench.py's launch paths.  `--gpus 2` without a launcher must start two ranks itself, see both through a real
collective and report the whole-job line with the PPO leg that carries the update collectives (rehearsed with gloo on the
one GPU of the test box: LG_BENCH_BACKEND=gloo; on an N-GPU node the same code runs with backend nccl = RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))


def _run(extra, env=None, expect_rc=0):
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra, capture_output=True, text=True, timeout=420, env=e, cwd=REPO)
    assert r.returncode == expect_rc, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_short_run_is_repeated_and_kernel_time_is_consistent():
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--training-iters", "0"])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["repeats"] >= 9
    assert out["roofline"]["kernel_ms"] <= out["ms_per_step"] * 1.0001          # events bracket the same replays inside the host bracket
    assert 0.0 < out["roofline"]["frac"] < 1.0 and 0.0 < out["roofline"]["flops"]["frac"] < 1.0
    assert out["config"]["state_finite"] and "self-collision" in out["config"]["workload"]


def test_gpus_2_spawns_two_ranks_and_reports_the_collectives():
    out = _run(["--gpus", "2", "--steps", "40", "--warmup", "20", "--num-envs", "1024", "--no-cpu-baseline", "--training-iters", "3"],
               env={"LG_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["config"]["parallelism"] == "env-sharded x2"
    tr = out["ppo_training"]
    assert "error" not in tr and tr["collectives_per_iteration"]["ranks"] == 2
    # kernel update path: ONE collective per mini-batch step (all gradients + the KL in one flat buffer), one all-gather per iteration
    assert tr["collectives_per_iteration"]["all_gather_returns_advantages"] == 1 and tr["collectives_per_iteration"]["flat_all_reduce_all_gradients_and_kl"] == 20
    assert tr["value"] > 0 and out["value"] > 0


@pytest.mark.parametrize("task,envs,contact", [("anymal_c_rough", 4096, "height-field contact"), ("cassie", 8192, "height-field contact")])
def test_bench_lines_of_configs_3_and_5(task, envs, contact):
    """BASELINE.json configs 3 and 5 through bench.py: default env counts (8192 for cassie), height-field contact unless --trimesh, the
    wide actor kernel in the graph, finite state, and a PPO leg on the wide learner kernels."""
    out = _run(["--task", task, "--steps", "40", "--warmup", "20", "--no-cpu-baseline", "--training-iters", "2"])
    assert out["config"]["envs_per_gpu"] == envs and contact in out["config"]["workload"] and out["config"]["state_finite"]
    assert "k_policy_act_wide" in out["config"]["policy"] and out["value"] > 1e7
    assert 0.0 < out["roofline"]["frac"] < 1.0 and out["roofline"]["algorithmic_bytes_per_env_step"] > 1000
    tr = out["ppo_training"]
    assert "error" not in tr and tr["value"] > 1e6 and "wide learner kernels" in tr["update_path"]


@pytest.mark.parametrize("fault", ["raise:1", "hang:1"])
def test_a_failing_ppo_leg_at_world_2_leaves_the_headline_line(fault):
    """The PPO leg holds the collectives of the N > 1 line; a rank that raises or stalls in it must not cost the headline: rank 0 still
    prints its ONE line, with the error recorded in `ppo_training`, and every rank exits (no rank left inside a collective) -- with a
    NON-ZERO code (3), so the driver does not book a run whose training leg hung as a success."""
    out = _run(["--gpus", "2", "--steps", "40", "--warmup", "20", "--num-envs", "512", "--no-cpu-baseline", "--training-iters", "2"],
               env={"LG_BENCH_BACKEND": "gloo", "LG_BENCH_PPO_FAULT": fault, "LG_BENCH_PPO_TIMEOUT_S": "20"}, expect_rc=3)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["gloo_ranks"] == 2
    assert "error" in out["ppo_training"]


def test_bench_under_the_drivers_launcher():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`: bench.py must take rank / world from the env (not spawn again) and rank 0 alone prints the line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, LG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20", "--num-envs", "512", "--no-cpu-baseline", "--training-iters", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=e, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gloo_ranks"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["ppo_training"]["collectives_per_iteration"]["ranks"] == 2 and "[optimiser HIP graph]" in out["ppo_training"]["update_path"]
    assert out["ppo_training"]["host_launches_per_update"] == {"graph_replays": 40, "collectives": 20, "kernel_launches_from_host": 0}


@pytest.mark.parametrize("task", ["anymal_c_rough", "cassie"])
def test_config_3_and_5_lines_state_what_was_timed(task):
    """`dtype` names the arithmetic of the kernels actually timed (split-bf16 actor unless --f32-actor), the CPU baseline
    runs on the GPU leg's own height field, and roofline.traffic comes from a rocprofv3 --pmc pass of that config (profiles/r03_pmc_summary.json)."""
    out = _run(["--task", task, "--steps", "40", "--warmup", "20", "--training-iters", "0", "--cpu-seconds", "1"])
    assert "bf16x3 actor" in out["dtype"] and out["dtype"].startswith("f32 physics")
    assert out["roofline"]["traffic"] is not None and out["roofline"]["traffic"] > 1e6 and "r03_pmc_summary.json" in out["roofline"]["traffic_source"]
    assert "int16 curriculum height field" in out["cpu_baseline"]["sample"] and "on the plane" not in out["cpu_baseline"]["sample"]
    assert out["cpu_baseline"]["value"] > 1e4
    f32 = _run(["--task", task, "--steps", "40", "--warmup", "20", "--training-iters", "0", "--no-cpu-baseline", "--f32-actor"])
    assert f32["dtype"] == "f32"


def test_headline_line_uses_the_rollout_kernel_and_its_own_counters():
    out = _run(["--steps", "40", "--warmup", "20", "--training-iters", "0", "--no-cpu-baseline"])
    r = out["roofline"]
    assert "lg_rollout_policy" in out["config"]["launch"] and r["policy_steps_per_launch"] == 20 and out["dtype"] == "f32"
    assert r["traffic"] is not None and abs(r["launch_ms"] - 20 * r["kernel_ms"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
