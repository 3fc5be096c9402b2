#!/usr/bin/env python3
"""Diagnostic: shader-cycle shares of a fused policy's phases from the -DONGYM_STAMPS build (see tools/diag_stamps.py).
    python tools/diag_stamps_policy.py POLICY [B] [steps]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
os.environ["ONGYM_HIP_LIB"] = os.path.join(REPO, "optical-networking-gym_amd", "csrc", "variants", "lib_stamps.so")
import numpy as np  # noqa: E402
import bench  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402

pid = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
wl = bench.WORKLOADS["nsfnet320"]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                      num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                      load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
env.seed(1); env.reset()
env.step_policy(800, record=False, policy=1)      # generic kernel (the stamped lean kernel uses other indices)
out = (C.c_ulonglong * 16)()
env.lib.ongym_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
env.lib.ongym_debug_stamps(env._h, out)
env.step_policy(steps, record=False, policy=pid); env.sync()
ms = env.last_kernel_ms()
env.lib.ongym_debug_stamps(env._h, out)
v = np.array(list(out)[:16], np.float64)
print(f"policy {pid}: kernel {ms:.2f} ms for {steps} steps of {B} replicas (stamped build)")
for i, x in enumerate(v):
    print(f"  stamp {i:2d} {100 * x / v.sum():5.1f} %   {x / (B * steps):12.0f} cycles/step/wave")
