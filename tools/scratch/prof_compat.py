import cProfile, pstats, os, sys, time
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import optical_networking_gym.heuristics.heuristics as H
from optical_networking_gym.topology import bundled_topology_path, get_topology
from optical_networking_gym.wrappers.qrmsa_gym import QRMSAEnvWrapper
import bench
topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, bench.jocn_modulations(), 80, 0.2, 4.5, 5)
env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=300, episode_length=1000,
                      num_spectrum_resources=320, launch_power_dbm=0.0, bandwidth=4e12, frequency_start=3e8 / 1565e-9,
                      frequency_slot_bandwidth=12.5e9, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
                      margin=0, file_name="", measure_disruptions=False, k_paths=5, modulations_to_consider=6,
                      defragmentation=False, n_defrag_services=0, gen_observation=False, sync_views=False)
env.reset()
def run(n):
    for _ in range(n):
        action, _, _ = H.heuristic_shortest_available_path_first_fit_best_modulation(env)
        _, _, done, _, info = env.step(action)
        if done:
            env.reset()
run(300)
t0=time.perf_counter(); run(3000); print("steps/s", 3000/(time.perf_counter()-t0))
pr = cProfile.Profile(); pr.enable(); run(3000); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
