#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused first-fit policy + step loop (graph_load.py:161-163 of the reference) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one batched env-step: every one of the B replicas on a GPU processes one service request (policy + step +
next-request generation + departures).  Weak scaling: every rank owns B replicas on its own GPU (one process per GPU);
the replicas are independent, so the only collective is the RCCL all-reduce of the statistics vector after the timed
region.  Inputs are generated on device (counter-based traffic stream, include/ongym_traffic.h) and the whole state is
resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), os.path.join(REPO, "tests")]

import numpy as np  # noqa: E402

WORKLOADS = {
    # BASELINE.json metric: QRMSA NSFNET-320, k=5, load 300 (SURVEY §8d synthetic inputs), JOCN modulation set
    "nsfnet320": dict(topology="nsfnet_chen.txt", S=320, load=300.0, capacity=448, bit_rates=(10, 40, 100, 400)),
    "cost239_320": dict(topology="cost239.txt", S=320, load=400.0, capacity=512, bit_rates=(10, 40, 100, 400)),
    "nobeleu768": dict(topology="nobel-eu.txt", S=768, load=600.0, capacity=704, bit_rates=(10, 40, 100, 400)),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def jocn_modulations():
    from optical_networking_gym.topology import Modulation
    return (Modulation("BPSK", 100000, 1, 3.71, -14), Modulation("QPSK", 2000, 2, 6.72, -17),
            Modulation("8QAM", 1000, 3, 10.84, -20), Modulation("16QAM", 500, 4, 13.24, -23),
            Modulation("32QAM", 250, 5, 16.16, -26), Modulation("64QAM", 125, 6, 19.01, -29))


def build_tables(name):
    from optical_networking_gym._tables import StaticTables
    from optical_networking_gym.topology import bundled_topology_path, get_topology
    topo = get_topology(bundled_topology_path(name), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    return StaticTables.from_topology(topo)


def algorithmic_bytes_per_step(stats_sum, S):
    """T*H*ceil(S/8) + T*N*4 + 2*H*16 + 24*log2(A+1) + 64 with the measured T, H, N, A."""
    steps = max(int(stats_sum["total_steps"]), 1)
    T = stats_sum["total_paths_tried"] / steps
    H = stats_sum["total_path_hops"] / max(stats_sum["total_paths_tried"], 1)
    G = stats_sum["total_gn_evals"] / steps
    N = stats_sum["total_interferer_terms"] / max(stats_sum["total_gn_evals"], 1)
    A = stats_sum["total_active_sum"] / steps
    b = T * H * math.ceil(S / 8) + T * N * 4 + 2 * H * 16 + 24 * math.log2(A + 1) + 64
    return b, dict(paths_tried_per_step=T, mean_hops=H, gn_evals_per_step=G, interferer_link_terms_per_gn=N,
                   mean_active_services=A)


def cpu_baseline(tables, wl, seconds_target=12.0):
    """The CPU oracle (C restatement of the reference, oracle/) on this box's host cores, bounded sample."""
    from optical_networking_gym import _native as nat
    from oracle_lib import OracleEnv, batch_run_first_fit
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    nrep = threads * 2
    holder = nat.ConfigHolder(tables, modulations=jocn_modulations(), num_spectrum_resources=wl["S"], batch=nrep,
                              capacity=wl["capacity"], episode_length=1000, auto_reset=True, load=wl["load"],
                              bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    envs = []
    for r in range(nrep):
        o = OracleEnv(holder, replica=r)
        o.seed(1)
        o.reset()
        envs.append(o)
    batch_run_first_fit(envs, 999, threads)          # warm-up episode (fills the network), untimed
    done, t0 = 0, time.perf_counter()
    while True:
        done += batch_run_first_fit(envs, 999, threads)
        dt = time.perf_counter() - t0
        if dt >= seconds_target or done >= 40 * 999 * nrep:
            break
    return dict(value=done / dt, unit="env-steps/s", cores=threads, kind="port",
                sample=f"{nrep} replicas x {done // nrep} steps of the same workload after a 999-step warm-up, "
                       f"OpenMP over replicas, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=65536, help="replicas per GPU")
    ap.add_argument("--workload", default="nsfnet320", choices=sorted(WORKLOADS))
    ap.add_argument("--steps-per-launch", type=int, default=250)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    from optical_networking_gym._dist import init_process_group, rank_seed, reduce_run_statistics
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank
    dist = init_process_group("nccl", local_rank) if (world > 1 or "RANK" in os.environ) else None   # "nccl" = RCCL

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist:
        dist.barrier()
    from optical_networking_gym.envs.batched import BatchedQRMSAEnv

    wl = WORKLOADS[args.workload]
    tables = build_tables(wl["topology"])
    env = BatchedQRMSAEnv(tables=tables, modulations=jocn_modulations(), batch_size=args.batch, device=local_rank,
                          num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000,
                          auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    env.seed(rank_seed(args.seed, rank))
    env.reset()

    def run(nsteps, timed):
        kernel_ms, launches = 0.0, 0
        left = nsteps
        while left > 0:
            n = min(left, args.steps_per_launch)
            env.step_policy(n, record=False)
            if timed:
                kernel_ms += env.last_kernel_ms()     # HIP events on the env's own stream
                launches += 1
            left -= n
        return kernel_ms, launches

    run(args.warmup, False)
    env.sync()
    s0 = env.stats()

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    kernel_ms, launches = run(args.steps, True)
    env.sync()
    fence()
    dt = time.perf_counter() - t0
    s1 = env.stats()

    fields = ("total_steps", "total_accepted", "total_gn_evals", "total_interferer_terms", "total_paths_tried",
              "total_path_hops", "total_active_sum")
    delta = np.array([float(s1[f].sum() - s0[f].sum()) for f in fields], np.float64)
    delta, dt_max, kernel_ms = reduce_run_statistics(delta, dt, kernel_ms, dist)   # the only collective (RCCL)
    stats_sum = dict(zip(fields, delta))
    expected = float(args.batch) * args.steps * world
    if int(stats_sum["total_steps"]) != int(expected):
        raise SystemExit(f"step accounting mismatch: {stats_sum['total_steps']} != {expected}")

    if rank == 0:
        value = expected / dt_max
        bytes_step, counters = algorithmic_bytes_per_step(stats_sum, wl["S"])
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        steps_per_launch_total = float(args.batch) * min(args.steps_per_launch, args.steps)
        achieved = bytes_step * steps_per_launch_total / avg_launch_s / 1e9 if launches else 0.0
        out = {
            "metric": "env-steps/s (requests/s), QRMSA NSFNET-320 batch=65k, 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"QRMSA {args.workload}: {tables.name} {tables.n_nodes}n/{tables.n_links}e, "
                                   f"S={wl['S']}, k=5, 6 modulations, load {wl['load']} Erlang, discrete bit rates "
                                   f"{wl['bit_rates']}, episode_length 1000 with auto-reset, fused first-fit policy+step",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "steps_per_launch": args.steps_per_launch, "parallelism": f"replica-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_run<uniform_alpha,rec32,waves> (csrc/ongym_hip.hip)", "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_env_step": bytes_step, "env_steps_per_launch": steps_per_launch_total,
                         **counters},
            "blocking_rate": 1.0 - stats_sum["total_accepted"] / stats_sum["total_steps"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tables, wl)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
