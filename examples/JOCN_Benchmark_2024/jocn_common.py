"""Shared pieces of the batched JOCN-benchmark drivers (reference: examples/JOCN_Benchmark_2024/graph_load.py,
graph_launch_power.py, graph_margin.py). One replica = one independent simulation; sweeps are a batch dimension."""
import os
import sys
import time
from datetime import datetime

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "optical-networking-gym_amd"))

from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402
from optical_networking_gym.topology import Modulation, bundled_topology_path, get_topology  # noqa: E402

BUNDLED = {"nobel-eu.xml": "nobel-eu.txt", "nobel-eu.txt": "nobel-eu.txt", "nsfnet_chen.txt": "nsfnet_chen.txt",
           "cost239.txt": "cost239.txt", "ring_4.txt": "ring_4.txt"}


def jocn_modulations():
    # reference graph_load.py:252-295
    return (Modulation("BPSK", 100_000, 1, 3.71, -14), Modulation("QPSK", 2_000, 2, 6.72, -17),
            Modulation("8QAM", 1_000, 3, 10.84, -20), Modulation("16QAM", 500, 4, 13.24, -23),
            Modulation("32QAM", 250, 5, 16.16, -26), Modulation("64QAM", 125, 6, 19.01, -29))


def load_topology(name, k_paths=5):
    path = name if os.path.exists(name) else bundled_topology_path(BUNDLED.get(name, name))
    return get_topology(path, None, jocn_modulations(), 80, 0.2, 4.5, k_paths)


def csv_header(modulations):
    # reference graph_load.py:144-155
    h = ("episode,service_blocking_rate,episode_service_blocking_rate,bit_rate_blocking_rate,"
         "episode_bit_rate_blocking_rate, episode_service_realocations, episode_defrag_cicles")
    for mf in modulations:
        h += f",modulation_{mf.spectral_efficiency}"
    return h + ",episode_disrupted_services,episode_time,mean_gsnr\n"


def csv_row(ep, st, n_mods, ep_time):
    # reference graph_load.py:169-186 (info of the terminal step + mean of Service.OSNR)
    row = (f"{ep},{st['last_service_blocking_rate']},{st['last_episode_service_blocking_rate']},"
           f"{st['last_bit_rate_blocking_rate']},{st['last_episode_bit_rate_blocking_rate']},"
           f"{int(st['last_episode_service_reallocations'])},{int(st['last_episode_defrag_cycles'])}")
    for m in range(n_mods):
        row += f",{int(st['last_modulation_hist'][m])}"
    # info["episode_disrupted_services"] divides two C ints in the reference (qrmsa.pyx:1038-1041)
    acc = int(st["last_episode_accepted"])
    disrupted = float(int(st["last_episode_disrupted"]) // acc) if acc > 0 and st["last_episode_disrupted"] > 0 else 0.0
    return row + f",{disrupted},{ep_time:.2f},{st['last_mean_gsnr']}\n"


def run_sweep(topology, *, n_episodes, episode_length, replicas_per_point, points, seed, common, monitor_names,
              policy=0):
    """points: list of dicts with per-point overrides among launch_power_dbm / load / margin.
    Every point gets `replicas_per_point` replicas; each replica runs ceil(n_episodes / replicas_per_point) episodes.
    Writes one CSV per point (monitor_names[i]) with the reference's columns; returns per-point arrays of
    episode_service_blocking_rate."""
    P, R = len(points), replicas_per_point
    B = P * R
    rep = lambda key, default: np.repeat([pt.get(key, default) for pt in points], R)  # noqa: E731
    env = BatchedQRMSAEnv(topology, batch_size=B, episode_length=episode_length, auto_reset=True,
                          replica_launch_power_dbm=rep("launch_power_dbm", common.get("launch_power_dbm", 0.0)),
                          replica_load=rep("load", common["load"]), replica_margin=rep("margin", common.get("margin", 0.0)),
                          **{k: v for k, v in common.items() if k not in ("launch_power_dbm", "margin")})
    env.seed(seed)
    env.reset()
    rounds = -(-n_episodes // R)
    mods = env.modulations
    files = []
    for name in monitor_names:
        os.makedirs(os.path.dirname(name) or ".", exist_ok=True)
        f = open(name, "wt", encoding="UTF-8")
        f.write(f"# Date: {datetime.now()}\n")
        f.write(csv_header(mods))
        files.append(f)
    blocking = [[] for _ in points]
    ep_counter = [0] * P
    for _ in range(rounds):
        t0 = time.time()
        env.step_policy(episode_length - 1, record=False, policy=policy)   # one episode of every replica (auto-reset at the end)
        st = env.stats()
        dt = time.time() - t0
        for p in range(P):
            for r in range(R):
                if ep_counter[p] >= n_episodes:
                    break
                s = st[p * R + r]
                files[p].write(csv_row(ep_counter[p], s, len(mods), dt))
                blocking[p].append(float(s["last_episode_service_blocking_rate"]))
                ep_counter[p] += 1
    for f in files:
        f.close()
    env.close()
    return [np.array(b) for b in blocking]


def run_sweep_plugin(topology, heuristic, *, n_episodes, episode_length, points, seed, common, monitor_names, k_paths=None):
    """The same sweep for a policy that exists only as a plugin (f(env) -> (action, flag, flag), e.g. heuristic 3 of the
    reference's graph_load.py): one device-backed QRMSAEnvWrapper per point, episodes in sequence, the reference's loop
    (graph_load.py:157-186) verbatim.  Orders of magnitude slower than the fused policies."""
    from optical_networking_gym.wrappers.qrmsa_gym import QRMSAEnvWrapper
    S = common.get("num_spectrum_resources", 320)
    if k_paths is None:     # the routes the topology was built with (the CLI's -k)
        k_paths = int(topology.graph.get("k_paths") or max(len(v) for v in topology.graph["ksp"].values()))
    blocking = []
    for i, (pt, name) in enumerate(zip(points, monitor_names)):
        kw = dict(common)
        kw.update(pt)
        kw.pop("capacity", None)
        env = QRMSAEnvWrapper(topology=topology, seed=seed + i, allow_rejection=True, episode_length=episode_length,
                              bandwidth=S * 12.5e9, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                              k_paths=k_paths, modulations_to_consider=6, gen_observation=False, **kw)
        env.reset()
        os.makedirs(os.path.dirname(name) or ".", exist_ok=True)
        rates = []
        with open(name, "wt", encoding="UTF-8") as f:
            f.write(f"# Date: {datetime.now()}\n")
            f.write(csv_header(env.env.modulations))
            for ep in range(n_episodes):
                env.reset()
                done, t0 = False, time.time()
                while not done:
                    action, _, _ = heuristic(env)
                    _, _, done, _, info = env.step(action)
                row = (f"{ep},{info['service_blocking_rate']},{info['episode_service_blocking_rate']},"
                       f"{info['bit_rate_blocking_rate']},{info['episode_bit_rate_blocking_rate']},"
                       f"{info['episode_service_realocations']},{info['episode_defrag_cicles']}")
                for mf in env.env.modulations:
                    row += f",{info.get(f'modulation_{float(mf.spectral_efficiency)}', 0)}"
                services = env.env.topology.graph["services"]
                mean_gsnr = sum(s.OSNR for s in services) / len(services) if services else 0.0
                f.write(row + f",{info.get('episode_disrupted_services', 0)},{time.time() - t0:.2f},{mean_gsnr}\n")
                rates.append(float(info["episode_service_blocking_rate"]))
        env.close()
        blocking.append(np.array(rates))
    return blocking
