"""VecEnv buffer contract (reference ``legged_gym/envs/base/base_task.py:38-121``).

Headless only: the viewer / keyboard half of the reference class
(``base_task.py:91-102, 123-147``) is out of scope.  There is no ``gym``
handle; the simulator is the HIP library behind ``DeviceSim``.
"""
import torch


def parse_device_str(dev: str):
    """Stand-in for gymutil.parse_device_str: 'cuda:1' -> ('cuda', 1)."""
    dev = str(dev)
    if ":" in dev:
        kind, idx = dev.split(":")
        return kind, int(idx)
    return dev, 0


class BaseTask:
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        sim_device_type, self.sim_device_id = parse_device_str(self.sim_device)
        self.headless = headless
        # reference: env tensors live on the sim device iff GPU pipeline (base_task.py:50-54).
        # The product path is GPU only; a CPU request fails loudly in DeviceSim.
        if sim_device_type in ("cuda", "gpu") and getattr(sim_params, "use_gpu_pipeline", True):
            self.device = f"cuda:{self.sim_device_id}"
        else:
            self.device = "cpu"
        self.graphics_device_id = -1 if headless else self.sim_device_id

        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions

        self.extras = {}
        self.viewer = None
        self.enable_viewer_sync = True
        # buffers (obs_buf, rew_buf, reset_buf, episode_length_buf, time_out_buf) are allocated by
        # create_sim() through DeviceSim so the kernels write straight into them (T4 of SURVEY 8a)
        self.create_sim()
        if self.num_privileged_obs is not None:
            self.privileged_obs_buf = torch.zeros(self.num_envs, self.num_privileged_obs, device=self.device, dtype=torch.float)
        else:
            self.privileged_obs_buf = None

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        """Reset all robots, then one zero-action step (base_task.py:114-118)."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(
            torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False))
        return obs, privileged_obs

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        return None     # headless
