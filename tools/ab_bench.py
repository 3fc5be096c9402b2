#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libongym_hip.so (one subprocess per build per round).

    python tools/ab_bench.py [--rounds 3] [--batch 65536] [--steps 250] libA.so libB.so ...
Prints per-build kernel ms per launch (min / median) and env-steps/s. Device time from HIP events only."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path[:0] = [os.path.join(%(repo)r, "optical-networking-gym_amd"), %(repo)r]
import bench
from optical_networking_gym.envs.batched import BatchedQRMSAEnv
wl = bench.WORKLOADS[%(workload)r]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(),
                      batch_size=%(batch)d, num_spectrum_resources=wl["S"], capacity=%(capacity)d or wl["capacity"], episode_length=1000,
                      auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
env.seed(1); env.reset()
env.step_policy(%(warm)d, record=False); env.sync()
ms = []
for _ in range(%(reps)d):
    env.step_policy(%(steps)d, record=False); env.sync(); ms.append(env.last_kernel_ms())
st = env.stats()
print(json.dumps(dict(ms=ms, acc=float(st["total_accepted"].sum()), steps=float(st["total_steps"].sum()))))
'''

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--reps", type=int, default=8, help="launches per round; a multiple of 4 covers whole episodes at 250 steps")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=250)
ap.add_argument("--warm", type=int, default=1000)
ap.add_argument("--workload", default="nsfnet320")
ap.add_argument("--capacity", type=int, default=0)
a = ap.parse_args()
res = {l: [] for l in a.libs}
check = {}
for rd in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ, ONGYM_HIP_LIB=os.path.abspath(lib))
        code = CHILD % dict(repo=REPO, workload=a.workload, batch=a.batch, warm=a.warm, reps=a.reps, steps=a.steps, capacity=a.capacity)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if out.returncode:
            print(lib, "FAILED", out.stderr[-800:]); continue
        r = json.loads(out.stdout.strip().splitlines()[-1])
        res[lib] += r["ms"]; check[lib] = (r["acc"], r["steps"])
for lib in a.libs:
    ms = sorted(res[lib])
    if not ms:
        continue
    med = ms[len(ms) // 2]
    mean = sum(ms) / len(ms)
    print(f"{os.path.basename(lib):40s} min {ms[0]:8.2f} ms  med {med:8.2f} ms  mean {mean:8.3f} ms -> {a.batch * a.steps / mean * 1e3:.4e} steps/s   accepted/steps {check[lib]}")
