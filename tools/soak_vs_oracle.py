#!/usr/bin/env python3
"""Soak: 1 024 replicas x 8 000 steps (eight episodes) per topology with launch power -8..+8 dBm, load 100..1000 Erlang and
margins 0..3 dB spread over the replicas; every replica's final grid, clocks and counters against the CPU oracle
(OpenMP over replicas, ~1 min).  Last run (round 2, lean kernel, profiles/r02_soak_vs_oracle.txt): 0 differing replicas of 1 024 on NSFNET and on COST239,
i.e. 16.4 M requests without one differing accept / slot decision.   python tools/soak_vs_oracle.py   (repository root, GPU)"""
import sys, os, time
sys.path[:0] = ["tests", "optical-networking-gym_amd"]
import numpy as np
from common import golden_tables, jocn_modulations
from oracle_lib import OracleEnv, batch_run_first_fit
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv
B, steps = 1024, 8000
rng = np.random.default_rng(7)
loads = rng.uniform(100, 1000, B); lps = rng.uniform(-8.0, 8.0, B); margins = rng.choice([0.0, 0.5, 1.5, 3.0], B)
for topo, S in (("nsfnet", 320), ("cost239", 320)):
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=1024, episode_length=1000,
              auto_reset=True, load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000),
              replica_load=loads, replica_launch_power_dbm=lps, replica_margin=margins)
    holder = nat.ConfigHolder(golden_tables(topo), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables(topo), modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=S,
                          capacity=1024, episode_length=1000, auto_reset=True, load=300, bit_rate_selection="discrete",
                          bit_rates=(10, 40, 100, 400, 1000), replica_load=loads, replica_launch_power_dbm=lps,
                          replica_margin=margins)
    env.seed(2025); env.reset()
    for _ in range(steps // 1000): env.step_policy(1000, record=False)
    st = env.stats()
    t0 = time.time()
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r); o.seed(2025); o.reset(); oracles.append(o)
    assert batch_run_first_fit(oracles, steps, 16) == B * steps
    bad = 0
    for r, o in enumerate(oracles):
        so = o.stats()
        ok = all(st[r][f] == so[f] for f in ("services_accepted", "bit_rate_provisioned", "current_time", "active", "rejected",
                                             "last_episode_accepted", "total_paths_tried")) and np.array_equal(env.grid(r), o.grid())
        bad += (not ok)
    print(topo, "replicas differing:", bad, "of", B, "| oracle time %.1f s" % (time.time() - t0), flush=True)
