#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused first-fit policy + step loop (graph_load.py:161-163 of the reference) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload NAME] [--scaling weak|strong]

A bench "step" is ONE LAUNCH of the hot path: every replica on the GPU processes `--steps-per-launch` (250) service
requests (policy + step + next-request generation + departures) with its state resident on the CU.  The metric stays
env-steps/s = requests/s summed over replicas and GPUs; `config.env_steps_per_bench_step` says how many a step holds.
Before anything is timed the network is always filled by >= one whole episode (999 env-steps per replica, untimed), so
the measured state does not depend on --warmup/--steps: the timed region runs on the steady-state episode mix of
`graph_load.py:157-164` (whole episodes with auto-reset), never on a cold network.  `--warmup W` then adds W untimed
bench steps and `--steps K` times exactly K of them between barrier + synchronize fences (max over ranks).

Multi-GPU: one process per GPU.  Under `python -m torch.distributed.run` the ranks come from the environment; a plain
`python bench.py --gpus N` spawns that launcher itself as a CHILD process before anything touches the GPU.  Replicas are
independent: rank k owns the global replicas [k*B, (k+1)*B) (request stream = (seed, global replica index), so a sharded
run simulates exactly the replicas of the unsharded one); the only collective is the RCCL all-reduce of the statistics
vector after the timed region.  `--scaling weak`: B = --batch per GPU; `--scaling strong`: --batch is the GLOBAL batch.
Inputs are generated on device (include/ongym_traffic.h) and all state is HBM-resident when the timed region starts.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction takes 2 cycles on its SIMD, 2.4 GHz max clock
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
# scalar ALU: one instruction per SIMD every ~4 cycles, whatever the number of waves (measured: tools/ubench/issue_rates.hip,
# profiles/r02_issue_rates_ubench.txt: 4.3 cycles per s_add_u32 per SIMD at 2..8 waves/SIMD)
SALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0
# reference (Cython) throughput measured in the build container, BASELINE.md §2 / SURVEY §6 (cannot travel to the GPU box)
REFERENCE_MEASURED = {
    "nsfnet320": {"steps_per_s_1core": 348.0, "steps_per_s_8procs": 2178.0},
    "nobeleu768": {"steps_per_s_1core_nsfnet768_load600": 282.0, "steps_per_s_8procs_nsfnet768_load600": 1636.0},
}
MIN_FILL_ENV_STEPS = 999     # one whole episode (warm-up 1 episode)


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


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.hpp, include/*.h, sorted by name) and the build flags: tools/pmc_summarise.py stores
    it with every counter summary, bench.py compares it with the sources it runs and marks a summary of other sources stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "optical-networking-gym_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) +
                    glob.glob(os.path.join(REPO, "include", "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    import __graft_entry__ as ge       # ... and the compiler flags of the build recipe (the lean units are built without machine LICM)
    h.update(" ".join(ge.HIP_FLAGS + ge.FAST_UNIT_FLAGS).encode())
    return h.hexdigest()[:16]


def pmc_key(workload, policy):
    return workload if policy == 0 else f"{workload}_p{policy}"


def pmc_summary(workload, policy=0):
    """Counter summary of the dominant kernel for this workload and policy, from the rocprofv3 --pmc passes committed under
    profiles/ (collected in separate runs as MI355X_MICROARCH.md prescribes; bench.py cannot run the profiler on
    itself).  None when no profile is tracked; `stale` is set when it was collected on other kernel sources."""
    path = os.path.join(REPO, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            summ = json.load(f).get(pmc_key(workload, policy))
    except (OSError, ValueError):
        return None
    if summ is not None:
        summ = dict(summ)
        summ["stale"] = summ.get("kernel_source_sha") != kernel_source_hash()
    return summ


def cpu_baseline(tables, wl, workload, policy=0, seconds_target=12.0):
    """The CPU oracle (C restatement of the reference, oracle/) on this box's host cores, bounded sample."""
    from optical_networking_gym import _native as nat
    from oracle_lib import OracleEnv, batch_run_first_fit, batch_run_policy
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
    if policy == 0:
        batch_run_first_fit(envs, 999, threads)          # warm-up episode (fills the network), untimed
        done, t0 = 0, time.perf_counter()
        while True:
            done += batch_run_first_fit(envs, 999, threads)
            dt = time.perf_counter() - t0
            if dt >= seconds_target or done >= 40 * 999 * nrep:
                break
        sample = (f"{nrep} replicas x {done // nrep} steps of the same workload after a 999-step warm-up, "
                  f"OpenMP over replicas, {dt:.1f} s")
    else:
        # the heuristics that evaluate many candidates run at tens of steps per second per core in the port: the network is
        # filled by 700 first-fit steps (untimed), then the policy runs in slices sized from the first one
        batch_run_first_fit(envs, 700, threads)
        t0 = time.perf_counter()
        done = batch_run_policy(envs, policy, 10, threads)
        dt = time.perf_counter() - t0
        chunk = int(max(10, min(250, 10 * (seconds_target - dt) / max(dt, 1e-3) / 4)))
        while dt < seconds_target and done < 290 * nrep:      # stays inside the first episode
            done += batch_run_policy(envs, policy, min(chunk, 290 - done // nrep), threads)
            dt = time.perf_counter() - t0
        sample = (f"{nrep} replicas x {done // nrep} steps of policy {policy} on the same workload after 700 first-fit steps "
                  f"(network filled), OpenMP over replicas, {dt:.1f} s")
    out = dict(value=done / dt, unit="env-steps/s", cores=threads, kind="port", sample=sample)
    if workload in REFERENCE_MEASURED and policy == 0:
        # the chain GPU -> port (timed here) -> reference (timed in the build container: it cannot travel)
        out["reference_measured"] = dict(REFERENCE_MEASURED[workload], unit="env-steps/s",
                                         where="build container, Xeon 2.1 GHz, 8 vCPU (BASELINE.md §2)")
    return out


def busy_frac(pmc, key, occ):
    cyc, wave = pmc.get(key), pmc.get("wave_cycles_per_env_step")
    if cyc is None or not wave or not occ.get("blocks_per_cu"):
        return None
    return cyc / (wave / (occ["blocks_per_cu"] / 4.0))


POLICY_NAMES = {0: "fused first-fit policy+step", 1: "fused load_balancing_best_modulation policy+step",
                2: "fused heuristic_highest_snr policy+step", 10: "fused heuristic_lowest_fragmentation policy+step"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a child process (nothing in this
    process has touched the GPU), stream its output through and exit with its code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed bench steps (launches of --steps-per-launch env-steps)")
    ap.add_argument("--warmup", type=int, default=4, help="untimed bench steps after the >= 1-episode fill")
    ap.add_argument("--batch", type=int, default=65536, help="replicas per GPU (weak) or in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--workload", default="nsfnet320", choices=sorted(WORKLOADS))
    ap.add_argument("--steps-per-launch", type=int, default=250)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", type=int, default=0,
                    help="fused policy id (include/ongym.h): 0 first fit (the BASELINE metric), 1 load balancing, 2 highest SNR, "
                         "10 lowest fragmentation have lean kernels; the others run the generic kernel")
    ap.add_argument("--record", action="store_true", help="write the per-step record (the reference's `info`) for every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps <= 0 or args.warmup < 0 or args.steps_per_launch <= 0:
        raise SystemExit("--steps and --steps-per-launch must be positive, --warmup non-negative")

    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # ONGYM_BENCH_REHEARSE=1: rehearsal of the N-rank path on a box with fewer GPUs than ranks (ranks share the devices
    # round-robin, the statistics reduction runs on gloo because RCCL refuses two ranks on one device). Never a result.
    rehearse = os.environ.get("ONGYM_BENCH_REHEARSE") == "1"
    device = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(device)
    from optical_networking_gym._dist import gather_per_rank, init_process_group, reduce_run_statistics, shard_bounds
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank
    backend = "gloo" if rehearse else "nccl"                                                          # "nccl" = RCCL
    dist = init_process_group(backend, device) if (world > 1 or "RANK" in os.environ) else None

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist:
        dist.barrier()
    from optical_networking_gym.envs.batched import BatchedQRMSAEnv

    wl = WORKLOADS[args.workload]
    spl = args.steps_per_launch
    global_batch = args.batch * world if args.scaling == "weak" else args.batch
    base, local_batch = shard_bounds(global_batch, rank, world)      # this rank's slice of the global replicas
    tables = build_tables(wl["topology"])
    env = BatchedQRMSAEnv(tables=tables, modulations=jocn_modulations(), batch_size=local_batch, device=device,
                          io_device=bool(args.record), num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000,
                          auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    env.seed(args.seed, replica_base=base)
    env.reset()

    rec_ptr = None
    if args.record:     # the per-step records (the reference's `info`, qrmsa.pyx:996-1060) of one launch, kept in HBM
        from optical_networking_gym import _native as nat
        rec_buf = torch.empty(spl * local_batch * nat.STEP_DTYPE.itemsize, dtype=torch.uint8, device=f"cuda:{device}")
        rec_ptr = rec_buf.data_ptr()

    def run(launches, timed):
        kernel_ms = 0.0
        for _ in range(launches):
            env.step_policy(spl, record=False, policy=args.policy, out_device_ptr=rec_ptr)
            if timed:
                kernel_ms += env.last_kernel_ms()     # HIP events on the env's own stream
        return kernel_ms

    fill_launches = -(-MIN_FILL_ENV_STEPS // spl)     # always: >= one whole episode, untimed
    run(fill_launches + args.warmup, False)
    env.sync()
    s0 = env.stats()

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    kernel_ms = run(args.steps, True)
    env.sync()
    fence()
    dt = time.perf_counter() - t0
    s1 = env.stats()

    fields = ("total_steps", "total_accepted", "total_gn_evals", "total_interferer_terms", "total_paths_tried",
              "total_path_hops", "total_active_sum")
    delta = np.array([float(s1[f].sum() - s0[f].sum()) for f in fields], np.float64)
    dt_rank, kernel_ms_rank = dt, kernel_ms
    delta, dt_max, kernel_ms = reduce_run_statistics(delta, dt, kernel_ms, dist,     # the only collective (RCCL)
                                                     device="cpu" if rehearse else "cuda")
    # what the process group saw: its size and every rank's own rate and launch time (one all_gather, after the timed region)
    per_rank = gather_per_rank([float(local_batch) * args.steps * spl / dt_rank, kernel_ms_rank / args.steps, float(local_batch)],
                               dist, device="cpu" if rehearse else "cuda")
    stats_sum = dict(zip(fields, delta))
    expected = float(global_batch) * args.steps * spl
    if int(stats_sum["total_steps"]) != int(expected):
        raise SystemExit(f"step accounting mismatch: {stats_sum['total_steps']} != {expected}")

    if rank == 0:
        value = expected / dt_max
        bytes_step, counters = algorithmic_bytes_per_step(stats_sum, wl["S"])
        avg_launch_s = kernel_ms / 1e3 / args.steps
        env_steps_per_launch = float(local_batch) * spl               # what ONE launch (on one GPU) processes
        achieved = bytes_step * env_steps_per_launch / avg_launch_s / 1e9
        pmc = pmc_summary(args.workload, args.policy)
        occ = env.occupancy(args.policy)
        traffic = None
        issue = None
        if pmc and pmc["stale"]:
            print(f"bench.py: profiles/pmc_summary.json[{pmc_key(args.workload, args.policy)}] was collected on other kernel "
                  f"sources ({pmc.get('kernel_source_sha')} != {kernel_source_hash()}): issue/traffic are marked stale",
                  file=sys.stderr)
        if pmc:
            hbm_b = pmc.get("hbm_bytes_per_env_step")
            if hbm_b is not None:
                traffic = hbm_b * env_steps_per_launch
            valu = pmc.get("valu_per_env_step")
            if valu is not None:
                rate = valu * (float(local_batch) * spl / avg_launch_s)
                salu = pmc.get("salu_per_env_step")
                salu_rate = salu * (float(local_batch) * spl / avg_launch_s) if salu is not None else None
                issue = {"bound": "instruction issue (scalar + vector ALU)", "valu_wave_insts_per_env_step": valu,
                         "salu_wave_insts_per_env_step": pmc.get("salu_per_env_step"),
                         "lds_wave_insts_per_env_step": pmc.get("lds_per_env_step"),
                         "smem_wave_insts_per_env_step": pmc.get("smem_per_env_step"),
                         "vmem_wave_insts_per_env_step": pmc.get("vmem_per_env_step"),
                         # SQ_ACTIVE_INST_VALU / _SCA (x4 clocks) against the SIMD-clocks one env-step takes:
                         # wave-cycles per env-step / resident waves per SIMD
                         "valu_busy_frac": busy_frac(pmc, "valu_active_cycles_per_env_step", occ),
                         "salu_busy_frac": busy_frac(pmc, "salu_active_cycles_per_env_step", occ),
                         "wait_any_frac": pmc.get("wait_any_frac"),
                         "stale": pmc["stale"], "kernel_source_sha": pmc.get("kernel_source_sha"),
                         "achieved": rate, "peak": VALU_ISSUE_PEAK, "unit": "VALU wave-insts/s",
                         "frac": rate / VALU_ISSUE_PEAK,
                         "salu_achieved": salu_rate, "salu_peak": SALU_ISSUE_PEAK,
                         "salu_frac": salu_rate / SALU_ISSUE_PEAK if salu_rate is not None else None,
                         "wave_cycles_per_env_step": pmc.get("wave_cycles_per_env_step"),
                         "source": pmc.get("source"), "note": "instruction counts per env-step from the tracked rocprofv3 "
                         "--pmc passes (separate runs); rate = counts x the env-step rate measured live in this run"}
        out = {
            "metric": "env-steps/s (requests/s), QRMSA NSFNET-320 batch=65k, 1/2/4/8 GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"QRMSA {args.workload}: {tables.name} {tables.n_nodes}n/{tables.n_links}e, "
                                   f"S={wl['S']}, k=5, 6 modulations, load {wl['load']} Erlang, discrete bit rates "
                                   f"{wl['bit_rates']}, episode_length 1000 with auto-reset, "
                                   f"{POLICY_NAMES.get(args.policy, f'fused policy {args.policy}+step')}"
                                   f"{', step records written' if args.record else ''}",
                       "policy": args.policy, "record": bool(args.record),
                       "batch_per_gpu": local_batch, "global_batch": global_batch,
                       "steps_per_launch": spl, "env_steps_per_bench_step": float(global_batch) * spl,
                       "fill_env_steps_per_replica_untimed": (fill_launches + args.warmup) * spl,
                       "parallelism": f"replica-sharded x{world}"},
            "roofline": {"bound": "hbm", "binding": "instruction issue (state is LDS-resident for a whole launch: see `issue`)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_stale": pmc["stale"] if (pmc and traffic is not None) else None,
                         "traffic_note": ("HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) per env-step from the "
                                          "tracked PMC passes x env-steps per launch; state is LDS-resident for a whole "
                                          "launch, so real traffic is far BELOW the algorithmic bytes and the HBM bound "
                                          "does not bind: see `issue`") if traffic is not None else None,
                         "kernel": (f"k_fast<M64,REC,ENT,WAVES,TRACE,POL={args.policy},WIDE> (csrc/ongym_fast.hpp)" if occ["lean_kernel"]
                                    else "k_run<uniform_alpha,codec,waves,policy> (csrc/ongym_device.hpp)"),
                         "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_env_step": bytes_step, "env_steps_per_launch": env_steps_per_launch,
                         **counters},
            "issue": issue,
            "blocking_rate": 1.0 - stats_sum["total_accepted"] / stats_sum["total_steps"],
        }
        # the timed state must be the loaded network, whatever the CLI said
        out["occupancy"] = occ
        # what the collective backend saw (n_gpus above comes from the launcher's WORLD_SIZE)
        out["process_group"] = {"world_size": dist.get_world_size() if dist else 1,
                                "backend": (dist.get_backend() if dist else None),
                                "per_rank_value": [r[0] for r in per_rank], "per_rank_avg_launch_ms": [r[1] for r in per_rank],
                                "per_rank_batch": [int(r[2]) for r in per_rank]}
        if rehearse:
            out["rehearsal"] = "ranks share GPUs, gloo reduction: plumbing check only, not a measurement"
        # the timed state must be the loaded network: by Little's law the carried load (offered x accepted fraction) is the mean
        # number of running services once the network is full; whole episodes from the empty network average ~0.7 of it
        carried = wl["load"] * (1.0 - out["blocking_rate"])
        out["steady_state"] = bool(counters["mean_active_services"] >= 0.4 * carried)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tables, wl, args.workload, args.policy)
        print(json.dumps(out), flush=True)
        if not out["steady_state"]:
            print("bench.py: mean active services far below the offered load: not the stated workload", file=sys.stderr)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
