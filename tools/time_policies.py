#!/usr/bin/env python3
"""Throughput of the fused policies other than first fit (whole batched episodes): tools/time_policies.py [B] [steps] [ids...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import bench  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ids = [int(a) for a in sys.argv[3:]] or list(range(12))
wl = bench.WORKLOADS["nsfnet320"]
for pid in ids:
    env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                          num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                          load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    env.seed(1); env.reset()
    env.step_policy(800, record=False); env.sync()            # near steady state with first fit (episodes are 1000 steps:
    n = steps if pid != 11 else max(steps // 10, 5)          # the timed steps stay inside the first one)
    assert 800 + n < 1000
    env.step_policy(n, record=False, policy=pid); env.sync()
    ms = env.last_kernel_ms()
    st = env.stats()
    print(f"policy {pid:2d}: B={B} {n} steps in {ms:9.2f} ms -> {B * n / ms * 1e3:.3e} env-steps/s; "
          f"mean active {st['active'].mean():.0f}", flush=True)
    env.close() if hasattr(env, "close") else None
