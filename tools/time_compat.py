#!/usr/bin/env python3
"""Steps/s of the single-environment compatibility surface (QRMSAEnvWrapper + the heuristics plugin API), i.e. what an
unmodified reference script sees: env.step() brings the next request, the statistics and the next decision of the
fused heuristic back in one call (ongym_step_actions_bundle); round 2: one launch + copy per heuristic call and per step.
    python tools/time_compat.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import optical_networking_gym.heuristics.heuristics as H  # noqa: E402
from optical_networking_gym.topology import bundled_topology_path, get_topology  # noqa: E402
from optical_networking_gym.wrappers.qrmsa_gym import QRMSAEnvWrapper  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for gen_obs, sync_views in ((False, True), (False, False), (True, False)):
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, bench.jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=300, episode_length=1000,
                          num_spectrum_resources=320, launch_power_dbm=0.0, bandwidth=4e12, frequency_start=3e8 / 1565e-9,
                          frequency_slot_bandwidth=12.5e9, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
                          margin=0, file_name="", measure_disruptions=False, k_paths=5, modulations_to_consider=6,
                          defragmentation=False, n_defrag_services=0, gen_observation=gen_obs, sync_views=sync_views)
    env.reset()
    t0 = time.perf_counter()
    n = steps if not gen_obs else steps // 4
    for _ in range(n):
        action, _, _ = H.heuristic_shortest_available_path_first_fit_best_modulation(env)
        _, _, done, _, info = env.step(action)
        if done:
            env.reset()
    dt = time.perf_counter() - t0
    print(f"gen_observation={gen_obs} sync_views={sync_views}: {n / dt:9.0f} env-steps/s "
          f"(heuristic + step through the gym surface; the reference: 348 steps/s on one core, 0.55 with observation)", flush=True)
