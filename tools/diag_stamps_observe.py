#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of k_observe from the -DONGYM_STAMPS build (never used for timing claims).
    hipcc ... -DONGYM_STAMPS -o csrc/libongym_hip_stamps.so ;  python tools/diag_stamps_observe.py"""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
os.environ["ONGYM_HIP_LIB"] = os.path.join(REPO, "optical-networking-gym_amd", "csrc", "variants", "lib_stamps.so")
import numpy as np  # noqa: E402
import bench  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402
NAMES = ["0 load_state", "1 path load/AND/blocks (to field)", "2 gn_build_list", "3 zero Fx + valid-start words", "4 needx marking",
         "5 xlist compaction", "6 tile prep (records, weights)", "7 field tiles (gathers)", "8 path setup + block stats", "9 (field tail)",
         "10 candidate compaction", "11 per-candidate GSNR/log10", "12 reductions", "13 feature finalise+store", "14 -", "15 tail"]
B = 16384
wl = bench.WORKLOADS["nsfnet320"]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                      num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                      load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
env.seed(1); env.reset(); env.step_policy(600, record=False)
out = (C.c_ulonglong * 16)()
env.lib.ongym_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
env.lib.ongym_debug_stamps(env._h, out)
env.observe(); ms = env.last_kernel_ms()
env.lib.ongym_debug_stamps(env._h, out)
v = np.array(list(out)[:16], np.float64); tot = v.sum()
print(f"k_observe (stamped) {ms:.3f} ms")
for n, x in zip(NAMES, v):
    print(f"  {n:36s} {100 * x / tot:5.1f} %   {x / B:9.0f} cycles/observation/wave")
print(f"  total {tot / B:.0f} cycles per observation per wave")
