"""Evaluation logger with the surface of reference ``legged_gym/utils/logger.py`` (``Logger(dt)``, ``log_state(s)``,
``log_rewards``, ``print_rewards``, ``plot_states``, ``reset``), used by ``scripts/play.py``.

Headless by construction: ``plot_states`` renders the nine panels of the reference (base velocities vs commands, one
joint's position / velocity / torque, foot forces, torque-velocity scatter) to a PNG with matplotlib's Agg backend, or,
when matplotlib is missing, dumps the logged series to a CSV next to it -- no window, no child process."""
import csv
import os
from collections import defaultdict

import numpy as np

# (row, col, title, xlabel, ylabel, [(series key, legend label)])
_PANELS = [
    (0, 0, "Base velocity x", "time [s]", "base lin vel [m/s]", [("base_vel_x", "measured"), ("command_x", "commanded")]),
    (0, 1, "Base velocity y", "time [s]", "base lin vel [m/s]", [("base_vel_y", "measured"), ("command_y", "commanded")]),
    (0, 2, "Base velocity yaw", "time [s]", "base ang vel [rad/s]", [("base_vel_yaw", "measured"), ("command_yaw", "commanded")]),
    (1, 0, "DOF Position", "time [s]", "Position [rad]", [("dof_pos", "measured"), ("dof_pos_target", "target")]),
    (1, 1, "Joint Velocity", "time [s]", "Velocity [rad/s]", [("dof_vel", "measured"), ("dof_vel_target", "target")]),
    (1, 2, "Base velocity z", "time [s]", "base lin vel [m/s]", [("base_vel_z", "measured")]),
    (2, 2, "Torque", "time [s]", "Joint Torque [Nm]", [("dof_torque", "measured")]),
]


class Logger:
    def __init__(self, dt, out_dir="."):
        self.dt, self.out_dir = float(dt), out_dir
        self.state_log, self.rew_log = defaultdict(list), defaultdict(list)
        self.num_episodes = 0

    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, values):
        for key, value in values.items():
            self.log_state(key, value)

    def log_rewards(self, episode_info, num_episodes):
        """``infos["episode"]`` holds per-second means of the episodes that just ended: weight them by their count."""
        for key, value in episode_info.items():
            if "rew" in key:
                self.rew_log[key].append(float(value) * num_episodes)
        self.num_episodes += num_episodes

    def reset(self):
        self.state_log.clear()
        self.rew_log.clear()

    def print_rewards(self):
        print("Average rewards per second:")
        for key, values in self.rew_log.items():
            print(f" - {key}: {np.sum(np.array(values)) / max(self.num_episodes, 1)}")
        print(f"Total number of episodes: {self.num_episodes}")

    def plot_states(self, filename="play_states.png"):
        """Write the state plots; returns the path of the file written (PNG, or CSV without matplotlib)."""
        log = self.state_log
        n = max((len(v) for v in log.values()), default=0)
        os.makedirs(self.out_dir, exist_ok=True)
        target = os.path.join(self.out_dir, filename)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            target = os.path.splitext(target)[0] + ".csv"
            scalars = [k for k, v in log.items() if v and np.ndim(v[0]) == 0]
            with open(target, "w", newline="") as fh:
                w = csv.writer(fh)
                w.writerow(["time"] + scalars)
                for i in range(n):
                    w.writerow([i * self.dt] + [log[k][i] if i < len(log[k]) else "" for k in scalars])
            return target
        t = np.linspace(0.0, n * self.dt, n)
        fig, axs = plt.subplots(3, 3, figsize=(15, 10))
        for r, c, title, xl, yl, series in _PANELS:
            for key, label in series:
                if log[key]:
                    axs[r, c].plot(t[:len(log[key])], log[key], label=label)
            axs[r, c].set(title=title, xlabel=xl, ylabel=yl)
        if log["contact_forces_z"]:
            f = np.array(log["contact_forces_z"])
            for i in range(f.shape[1]):
                axs[2, 0].plot(t[:len(f)], f[:, i], label=f"force {i}")
        axs[2, 0].set(title="Vertical Contact forces", xlabel="time [s]", ylabel="Forces z [N]")
        if log["dof_vel"] and log["dof_torque"]:
            axs[2, 1].plot(log["dof_vel"], log["dof_torque"], "x", label="measured")
        axs[2, 1].set(title="Torque/velocity curves", xlabel="Joint vel [rad/s]", ylabel="Joint Torque [Nm]")
        for a in axs.flat:
            if a.get_legend_handles_labels()[0]:
                a.legend()
        fig.tight_layout()
        fig.savefig(target, dpi=80)
        plt.close(fig)
        return target
