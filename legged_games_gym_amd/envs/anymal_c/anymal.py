"""ANYmal-C task: LeggedRobot whose torques come from the ANYdrive LSTM actuator network.

Reference ``legged_gym/envs/anymal_c/anymal.py:46-81``.  The network itself
(``torch.jit.load`` + one ``aten::lstm`` call per sub-step in the reference)
is evaluated inside the fused step kernel from the extracted fp32 weights; this
class only exposes the actuator state under the reference's names.
"""
from legged_games_gym_amd.envs.base.legged_robot import LeggedRobot


class Anymal(LeggedRobot):
    def _init_buffers(self):
        super()._init_buffers()
        if self.cfg.control.use_actuator_network:
            b = self._sim.buf
            self.sea_hidden_state = b["sea_hidden_state"]                     # [2, N*12, 8]
            self.sea_cell_state = b["sea_cell_state"]
            self.sea_hidden_state_per_env = self.sea_hidden_state.view(2, self.num_envs, self.num_actions, 8)
            self.sea_cell_state_per_env = self.sea_cell_state.view(2, self.num_envs, self.num_actions, 8)
    # reset_idx: the kernel zeroes the hidden / cell rows of reset envs (anymal.py:56-60)
    # _compute_torques: fused (anymal.py:71-81); PD fallback when use_actuator_network is False
