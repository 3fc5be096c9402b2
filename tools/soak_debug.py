#!/usr/bin/env python3
"""Diagnostic for a replica the soak reports as differing: tools/soak_debug.py CASES... (the case list of the soak run, the LAST one
is examined; --narrow as in the soak).  Prints the first step at which the device's step records leave the oracle's."""
import sys, os
sys.path[:0] = ["tests", "optical-networking-gym_amd"]
import numpy as np
from common import golden_tables, jocn_modulations
from oracle_lib import OracleEnv, batch_run_policy
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv

CASES = [("nsfnet", 320, 0, 1024, 8000), ("cost239", 320, 0, 1024, 8000), ("nobel-eu", 320, 0, 1024, 4000),
         ("nsfnet", 320, 1, 1024, 4000), ("nsfnet", 160, 2, 128, 1500), ("nsfnet", 160, 10, 128, 1500),
         ("nobel-eu", 320, 1, 512, 3000), ("nobel-eu", 160, 2, 96, 1200), ("nobel-eu", 160, 10, 96, 1200),
         ("cost239", 160, 2, 96, 1200), ("cost239", 160, 10, 96, 1200)]
NARROW = "--narrow" in sys.argv
BIT_RATES = (10, 40, 100, 400) if NARROW else (10, 40, 100, 400, 1000)
idx = [int(a) for a in sys.argv[1:] if a != "--narrow"]
rng = np.random.default_rng(7)
for ci in idx:       # the soak's draws, in its order
    topo, S, pid, B, steps = CASES[ci]
    loads = rng.uniform(100, 1000, B) * S / 320; lps = rng.uniform(-8.0, 8.0, B); margins = rng.choice([0.0, 0.5, 1.5, 3.0], B)
kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=1024, episode_length=1000, auto_reset=True, load=300,
          bit_rate_selection="discrete", bit_rates=BIT_RATES, replica_load=loads, replica_launch_power_dbm=lps, replica_margin=margins)
holder = nat.ConfigHolder(golden_tables(topo), batch=B, **kw)


def run_gpu():
    env = BatchedQRMSAEnv(tables=golden_tables(topo), batch_size=B, **kw)
    env.seed(2025); env.reset()
    recs, done = [], 0
    while done < steps:
        n = min(500, steps - done)
        recs.append(env.step_policy(n, record=True, policy=pid)); done += n
    return env, np.concatenate(recs)


env, rec = run_gpu()
print("lean kernel:", env.occupancy(pid), flush=True)
st = env.stats()
oracles = []
for r in range(B):
    o = OracleEnv(holder, replica=r); o.seed(2025); o.reset(); oracles.append(o)
batch_run_policy(oracles, pid, steps, len(os.sched_getaffinity(0)))
bad = []
for r, o in enumerate(oracles):
    so = o.stats()
    ok = all(st[r][f] == so[f] for f in ("services_accepted", "bit_rate_provisioned", "current_time", "active", "rejected")) and np.array_equal(env.grid(r), o.grid())
    if not ok:
        bad.append(r)
print("differing replicas:", bad, flush=True)
os.environ["ONGYM_FORCE_GENERIC"] = "1"
envg, recg = run_gpu()
print("generic kernel:", envg.occupancy(pid), flush=True)
for r in bad:
    o = OracleEnv(holder, replica=r); o.seed(2025); o.reset()
    ro = o.run_policy(pid, steps)
    print(f"replica {r}: load {loads[r]:.1f} launch power {lps[r]:.2f} dBm margin {margins[r]}")
    for name, g in (("lean", rec[:, r]), ("generic", recg[:, r])):
        first = None
        for t in range(steps):
            if any(g[f][t] != ro[f][t] for f in ("action", "accepted", "flags", "active", "nslots", "slot", "route", "modulation")):
                first = t; break
        print(f"  {name}: first differing step {first}")
        if first is not None:
            for t in range(max(first - 1, 0), min(first + 2, steps)):
                print(f"    step {t}: device {g[t]}\n    step {t}: oracle {ro[t]}")
