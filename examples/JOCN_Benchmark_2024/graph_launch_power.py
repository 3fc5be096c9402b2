#!/usr/bin/env python3
"""Launch-power sweep of the JOCN benchmark (Sec. 4.A of the paper; reference: examples/JOCN_Benchmark_2024/
graph_launch_power.py), batched: the 17 launch powers x R parallel simulations are replicas of ONE device environment.
Heuristic indices are the reference's (graph_launch_power.py:106-128); all of them are fused on device.

    python examples/JOCN_Benchmark_2024/graph_launch_power.py -t nobel-eu.xml -e 1000 -s 1000 -l 200
"""
import argparse

import numpy as np

from jocn_common import load_topology, run_sweep

# reference index -> fused device policy id (include/ongym.h)
FUSED = {1: 0, 2: 3, 3: 5, 4: 1, 5: 11, 6: 6, 7: 7, 8: 4, 9: 8, 10: 8}


def main():
    ap = argparse.ArgumentParser(description="Optical Network Simulation - launch power sweep (batched on GPU)")
    ap.add_argument("-t", "--topology_file", default="nobel-eu.xml")
    ap.add_argument("-e", "--num_episodes", type=int, default=100)
    ap.add_argument("-s", "--episode_length", type=int, default=1000)
    ap.add_argument("-l", "--load", type=float, default=200)
    ap.add_argument("-th", "--threads", type=int, default=64, help="parallel simulations (replicas) per launch power")
    ap.add_argument("-k", "--k_paths", type=int, default=5)
    ap.add_argument("-mf", "--monitor_file_name", default="examples/JOCN_Benchmark_2024/results/simulation_results")
    ap.add_argument("-hi", "--heuristic_index", type=int, default=1, choices=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10],
                    help="1 first fit, 2 lowest spectrum, 3 best-modulation load balancing, 4 load balancing best "
                         "modulation, 5 MSCL, 6 MSCL simplified, 7 MSCL sequential, 8 load-balancing first fit, "
                         "9 PSR-C, 10 PSR-O")
    ap.add_argument("--slots", type=int, default=320)
    ap.add_argument("--seed", type=int, default=20)
    args = ap.parse_args()

    topology = load_topology(args.topology_file, args.k_paths)
    launch_powers = np.linspace(-8, 8, num=17)
    common = dict(load=float(args.load), num_spectrum_resources=args.slots, bit_rate_selection="discrete",
                  bit_rates=(10, 40, 100, 400), capacity=1024)
    names = [f"{args.monitor_file_name}_{topology.graph['name']}_{lp}_{float(args.load)}.csv" for lp in launch_powers]
    points = [dict(launch_power_dbm=float(lp)) for lp in launch_powers]
    res = run_sweep(topology, n_episodes=args.num_episodes, episode_length=args.episode_length,
                    replicas_per_point=min(args.threads, args.num_episodes), seed=args.seed, common=common,
                    points=points, monitor_names=names, policy=FUSED[args.heuristic_index])
    for lp, b in zip(launch_powers, res):
        print(f"Launch power: {lp:.1f} dBm, mean: {b.mean():.4f}, stdev: {b.std(ddof=1) if len(b) > 1 else 0:.4f}")


if __name__ == "__main__":
    main()
