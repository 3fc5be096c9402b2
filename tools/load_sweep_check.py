#!/usr/bin/env python3
"""Informational: batched nobel-eu load sweep (first fit, +1 dBm, rates 10/40/100/400) next to the means printed in the
reference's plots.ipynb cell 32 (5 episodes per load, produced by an unknown earlier revision of the reference).
Above ~500 Erlang the two agree within the notebook's noise; below it the notebook is higher than what the present
reference code gives (which the oracle and the kernels reproduce bit for bit) - not used as a parity pin.

    python tools/load_sweep_check.py        (run from the repository root, needs the GPU)
"""
import sys, os
sys.path.insert(0, "examples/JOCN_Benchmark_2024")
import numpy as np, jocn_common as J
topology = J.load_topology("nobel-eu.xml", 5)
loads = np.arange(200, 1000, 100)
common = dict(load=200.0, num_spectrum_resources=320, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
              launch_power_dbm=1.0, capacity=1024)
names = [f"gpurun_out/ls/load_{ld}.csv" for ld in loads]
res = J.run_sweep(topology, n_episodes=512, episode_length=1000, replicas_per_point=256, seed=7, common=common,
                  points=[dict(load=float(ld)) for ld in loads], monitor_names=names, policy=0)
pub = [0.002202, 0.013814, 0.037037, 0.051852, 0.064665, 0.068268, 0.084885, 0.107307]
for ld, b, p in zip(loads, res, pub):
    print(f"load {ld}: mean {b.mean():.4f} (std over episodes {b.std():.4f})  published(5 episodes) {p:.4f}")
