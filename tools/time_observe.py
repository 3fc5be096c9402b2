#!/usr/bin/env python3
"""Device time of ongym_observe (observation + action mask of every replica) and of an RL-style step(actions)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import numpy as np  # noqa: E402
import bench  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402

batches = [int(a) for a in sys.argv[1:]] or [4096, 16384]
for B in batches:
    wl = bench.WORKLOADS["nsfnet320"]
    env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                          num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                          load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    env.seed(1); env.reset()
    env.step_policy(600, record=False)
    ms, ms_pol, ms_step = [], [], []
    for _ in range(5):
        obs, mask = env.observe()
        ms.append(env.last_kernel_ms())
        acts, _ = env.policy_actions()
        ms_pol.append(env.last_kernel_ms())
        env.step(acts)
        ms_step.append(env.last_kernel_ms())
    print(f"B={B}: observe kernel {np.median(ms):.3f} ms -> {B / np.median(ms) * 1e3:.3e} observations/s; "
          f"mask ones/replica {mask[:, :-1].sum() / B:.0f}; policy_actions kernel {np.median(ms_pol):.3f} ms, "
          f"step(actions) kernel {np.median(ms_step):.3f} ms")
