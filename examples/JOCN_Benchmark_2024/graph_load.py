#!/usr/bin/env python3
"""Load sweep of the JOCN benchmark (reference: examples/JOCN_Benchmark_2024/graph_load.py), batched: all loads x R
parallel simulations are replicas of ONE device environment; heuristics 1 (first fit), 2 (highest SNR),
3 (lowest fragmentation) and 4 (load balancing best modulation) are all fused on device.

    python examples/JOCN_Benchmark_2024/graph_load.py -t nobel-eu.xml -e 1000 -s 1000
"""
import argparse

import numpy as np

from jocn_common import load_topology, run_sweep


def get_loads(topology_name: str) -> np.ndarray:   # reference graph_load.py:17-29
    table = {"nobel-eu.xml": np.arange(100, 501, 100), "germany50.xml": np.arange(300, 801, 50),
             "janos-us.xml": np.arange(100, 601, 50), "nsfnet_chen.txt": np.arange(100, 601, 50),
             "ring_4.txt": np.arange(100, 601, 50), "cost239.txt": np.arange(100, 601, 50)}
    if topology_name not in table:
        raise ValueError(f"Unknown topology name: {topology_name}")
    return table[topology_name]


def main():
    ap = argparse.ArgumentParser(description="Optical Network Simulation - load sweep (batched on GPU)")
    ap.add_argument("-t", "--topology_file", default="nobel-eu.xml")
    ap.add_argument("-e", "--num_episodes", type=int, default=25)
    ap.add_argument("-s", "--episode_length", type=int, default=1000)
    ap.add_argument("-th", "--threads", type=int, default=25, help="parallel simulations (replicas) per load")
    ap.add_argument("-hi", "--heuristic_index", type=int, default=1, choices=[1, 2, 3, 4],
                    help="1: first fit, 2: highest SNR, 3: lowest fragmentation, 4: load balancing best modulation "
                         "(all fused on device; 3 rejects and flags the requests on which the reference raises its QoT "
                         "ValueError, qrmsa.pyx:925-929)")
    ap.add_argument("-df", "--defragmentation", action="store_true", help="defragment after departures (qrmsa.pyx:1117-1119)")
    ap.add_argument("-nd", "--n_defrag_services", type=int, default=0)
    ap.add_argument("-mf", "--monitor_file_name", default="examples/JOCN_Benchmark_2024/results/load_episodes")
    ap.add_argument("--launch_power", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=50)
    args = ap.parse_args()

    topology = load_topology(args.topology_file, 5)
    loads = get_loads(args.topology_file)
    common = dict(load=float(loads[0]), num_spectrum_resources=320, bit_rate_selection="discrete",
                  bit_rates=(10, 40, 100, 400, 1000), launch_power_dbm=args.launch_power, capacity=1024,
                  defragmentation=args.defragmentation, n_defrag_services=args.n_defrag_services)
    names = [f"{args.monitor_file_name}_{args.heuristic_index}_{topology.graph['name']}_{args.launch_power}_{float(ld)}.csv"
             for ld in loads]
    res = run_sweep(topology, n_episodes=args.num_episodes, episode_length=args.episode_length,
                    replicas_per_point=min(args.threads, args.num_episodes), seed=args.seed, common=common,
                    points=[dict(load=float(ld)) for ld in loads], monitor_names=names,
                    policy={1: 0, 2: 2, 3: 10, 4: 1}[args.heuristic_index])
    for ld, b in zip(loads, res):
        print(f"Load: {ld} Erlang, episode_service_blocking_rate mean: {b.mean():.4f}")


if __name__ == "__main__":
    main()
