"""Play-time state logger with the reference's interface (utils/logger.py:37-138).  Plotting needs matplotlib,
which is optional; without it plot_states() reports what it would have drawn."""
from collections import defaultdict

import numpy as np


class Logger:
    def __init__(self, dt):
        self.dt = dt
        self.reset()
        self.plot_process = None

    def reset(self):
        self.state_log = defaultdict(list)
        self.rew_log = defaultdict(list)
        self.num_episodes = 0

    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, values):
        for k, v in values.items():
            self.log_state(k, v)

    def log_rewards(self, values, num_episodes):
        for k, v in values.items():
            if "rew" in k:
                self.rew_log[k].append(float(v) * num_episodes)
        self.num_episodes += num_episodes

    def print_rewards(self):
        print("Average rewards per second:")
        for k, vals in self.rew_log.items():
            print(" - %s: %s" % (k, np.sum(np.array(vals)) / max(self.num_episodes, 1)))
        print("Total number of episodes: %d" % self.num_episodes)

    def plot_states(self):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            print("matplotlib unavailable; logged keys: %s" % sorted(self.state_log))
            return
        keys = sorted(self.state_log)
        fig, axes = plt.subplots(max(1, (len(keys) + 2) // 3), 3, squeeze=False)
        for ax, k in zip(axes.flat, keys):
            v = np.asarray(self.state_log[k], dtype=np.float64)
            ax.plot(np.arange(len(v)) * self.dt, v)
            ax.set_title(k)
        fig.savefig("play_states.png")
