#!/usr/bin/env python3
"""Diagnostic A/B: cost of the on-device request draw = throughput with the device generator vs replaying a
pre-generated trace of the same distribution (trace replay costs one 16-byte load per request)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import numpy as np
import bench
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = bench.WORKLOADS["nsfnet320"]
def make():
    return BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(),
                           batch_size=B, num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000,
                           auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
def timed(env, launches=4, steps=250):
    env.step_policy(1000, record=False); env.sync()
    ms = 0.0
    for _ in range(launches):
        env.step_policy(steps, record=False); env.sync(); ms += env.last_kernel_ms()
    return B * steps * launches / (ms * 1e-3)
env = make(); env.seed(1); env.reset()
print("device generator: %.3e steps/s" % timed(env)); env.close()
n = 1000 + 4 * 250 + 8
rng = np.random.default_rng(0)
reqs = np.zeros((B, n), nat.REQUEST_DTYPE)
iat = rng.exponential(10800.0 / wl["load"], (B, n)).astype(np.float32)
reqs["arrival_time"] = np.cumsum(iat, axis=1, dtype=np.float32)
reqs["holding_time"] = rng.exponential(10800.0, (B, n)).astype(np.float32)
src = rng.integers(0, 14, (B, n)); dst = (src + rng.integers(1, 14, (B, n))) % 14
reqs["source"] = src; reqs["destination"] = dst
reqs["bit_rate"] = np.asarray(wl["bit_rates"], np.float32)[rng.integers(0, len(wl["bit_rates"]), (B, n))]
env = make(); env.set_requests(reqs); env.reset()
print("trace replay    : %.3e steps/s" % timed(env))
