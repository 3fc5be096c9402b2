#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of k_run from the -DONGYM_STAMPS build (never used for timing claims).

    python __graft_entry__.py --variant stamps -DONGYM_STAMPS        (-> csrc/variants/lib_stamps.so)
    python tools/diag_stamps.py [--batch B] [--steps K] [--policy ID]
"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
os.environ["ONGYM_HIP_LIB"] = os.path.join(REPO, "optical-networking-gym_amd", "csrc", "variants", "lib_stamps.so")

import numpy as np  # noqa: E402
import bench  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402

FAST_NAMES = ["0 path record + bound prefilter", "1 path AND", "2 modulation loop: run-AND + first_set", "3 interferer list build",
              "4 interferer cache (prep)", "5 GN evaluation (gather, sum, test)", "6 accept: mark, record, counters / reject flags",
              "7 pop request (+ refill)", "8 release scan", "9 departures", "10 record / terminal", "11 load_state", "12 store_state",
              "13 route load (LB) / route score (LF)", "14 candidate compaction (HSNR, LF)", "15 candidates x interferers (HSNR, LF)"]
NAMES = ["0 request+nslots", "1 path load+AND", "2 run_and/first_set", "3 gn_build_list", "4 gn_eval", "5 mark_links",
         "6 lane0 bookkeeping+draw", "7 release_due", "8 load_state", "9 store_state",
         "10 gn: self term/setup", "11 gn: list+record LDS", "12 gn: pair-table gather", "13 gn: link-weight loop",
         "14 gn: wave_sum", "15 gn: tail (coef loads)"]

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=250)
ap.add_argument("--workload", default="nsfnet320")
ap.add_argument("--policy", type=int, default=0)
ap.add_argument("--warm", type=int, default=700)
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(),
                      batch_size=args.batch, num_spectrum_resources=wl["S"], capacity=wl["capacity"],
                      episode_length=1000, auto_reset=True, load=wl["load"], bit_rate_selection="discrete",
                      bit_rates=wl["bit_rates"])
env.seed(1); env.reset()
env.step_policy(args.warm, record=False)      # inside the first episode: the timed steps run on a loaded network
out = (C.c_ulonglong * 16)()
env.lib.ongym_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
env.lib.ongym_debug_stamps(env._h, out)
st0 = env.stats()
env.step_policy(args.steps, record=False, policy=args.policy)
env.sync()
ms = env.last_kernel_ms()
env.lib.ongym_debug_stamps(env._h, out)
v = np.array(list(out)[:16], np.float64)
tot = v.sum()
print(f"policy {args.policy}: kernel {ms:.2f} ms; {args.batch * args.steps / ms * 1e3:.3e} steps/s (stamped build), {env.occupancy(args.policy)}")
st = env.stats()
d = {f: float(st[f].sum() - st0[f].sum()) for f in ("total_paths_tried", "total_steps", "total_gn_evals", "total_interferer_terms", "total_accepted")}
print("  per step: paths tried %.2f, GN evaluations %.2f, interferer-link terms per evaluation %.1f, accepted %.4f" % (d["total_paths_tried"] / d["total_steps"], d["total_gn_evals"] / d["total_steps"], d["total_interferer_terms"] / max(d["total_gn_evals"], 1), d["total_accepted"] / d["total_steps"]))
for n, x in zip(FAST_NAMES if env.occupancy(args.policy)["lean_kernel"] else NAMES, v):
    print(f"  {n:28s} {100 * x / tot:5.1f} %   {x / (args.batch * args.steps):9.0f} cycles/step/wave")
print(f"  total {tot / (args.batch * args.steps):.0f} cycles per env-step per wave")
