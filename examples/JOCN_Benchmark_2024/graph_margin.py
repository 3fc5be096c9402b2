#!/usr/bin/env python3
"""Margin sweep of the JOCN benchmark (reference: examples/JOCN_Benchmark_2024/graph_margin.py, margins 0..2 dB in
0.5 dB steps), batched: every margin x R parallel simulations are replicas of ONE device environment.

    python examples/JOCN_Benchmark_2024/graph_margin.py -t nobel-eu.xml -e 1000 -s 1000 -l 210
"""
import argparse

import numpy as np

from jocn_common import load_topology, run_sweep


def main():
    ap = argparse.ArgumentParser(description="Optical Network Simulation - margin sweep (batched on GPU)")
    ap.add_argument("-t", "--topology_file", default="nsfnet_chen.txt")
    ap.add_argument("-e", "--num_episodes", type=int, default=1)
    ap.add_argument("-l", "--load", type=float, default=210)
    ap.add_argument("-s", "--episode_length", type=int, default=1000)
    ap.add_argument("-th", "--threads", type=int, default=64, help="parallel simulations (replicas) per margin")
    ap.add_argument("-hi", "--heuristic_index", type=int, default=1, choices=[1, 2, 4],
                    help="1: first fit, 2: highest SNR, 4: load balancing best modulation (graph_load.py numbering)")
    ap.add_argument("-mf", "--monitor_file_name", default="examples/JOCN_Benchmark_2024/results/mr_episodes")
    ap.add_argument("--launch_power", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=20)
    args = ap.parse_args()

    topology = load_topology(args.topology_file, 5)
    margins = np.arange(0, 2.1, 0.5)                       # reference graph_margin.py:146
    common = dict(load=float(args.load), num_spectrum_resources=320, bit_rate_selection="discrete",
                  bit_rates=(10, 40, 100, 400), launch_power_dbm=args.launch_power, capacity=1024)
    names = [f"{args.monitor_file_name}_{args.heuristic_index}_{mg}_{topology.graph['name']}_{args.launch_power}_"
             f"{float(args.load)}.csv" for mg in margins]
    res = run_sweep(topology, n_episodes=args.num_episodes, episode_length=args.episode_length,
                    replicas_per_point=min(args.threads, args.num_episodes), seed=args.seed, common=common,
                    points=[dict(margin=float(mg)) for mg in margins], monitor_names=names,
                    policy={1: 0, 2: 2, 4: 1}[args.heuristic_index])
    for mg, b in zip(margins, res):
        print(f"Margin: {mg:.1f} dB, episode_service_blocking_rate mean: {b.mean():.4f}")


if __name__ == "__main__":
    main()
