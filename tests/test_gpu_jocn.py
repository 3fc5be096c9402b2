"""Batched JOCN drivers: the launch-power sweep reproduces the blocking-vs-launch-power curve the
reference PUBLISHES (examples/JOCN_Benchmark_2024/plots.ipynb cell 12 output: nobel-eu, load 200, first fit, 17 launch
powers) — a statistical end-to-end check of policy + GN model + traffic, independent of the oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "examples", "JOCN_Benchmark_2024"))

# plots.ipynb:295 (mean, stdev of episode_service_blocking_rate per launch power -8..+8 dBm)
PUBLISHED_MEAN = [0.1263, 0.0906, 0.0464, 0.0258, 0.0095, 0.0036, 0.0011, 0.0009, 0.0007, 0.0005, 0.0008, 0.0011,
                  0.0042, 0.0414, 0.1225, 0.1838, 0.2344]
PUBLISHED_STD = [0.0077, 0.0098, 0.0051, 0.0065, 0.0037, 0.0024, 0.0013, 0.0008, 0.0009, 0.0007, 0.0011, 0.0012,
                 0.0027, 0.0072, 0.0071, 0.0111, 0.0156]


def test_launch_power_sweep_matches_published_curve(tmp_path):
    from jocn_common import load_topology, run_sweep
    topology = load_topology("nobel-eu.xml", 5)
    powers = np.linspace(-8, 8, 17)
    names = [str(tmp_path / f"lp_{p}.csv") for p in powers]
    res = run_sweep(topology, n_episodes=256, episode_length=1000, replicas_per_point=128, seed=20,
                    common=dict(load=200.0, num_spectrum_resources=320, bit_rate_selection="discrete",
                                bit_rates=(10, 40, 100, 400), capacity=1024),
                    points=[dict(launch_power_dbm=float(p)) for p in powers], monitor_names=names)
    means = np.array([r.mean() for r in res])
    for m, pm, ps in zip(means, PUBLISHED_MEAN, PUBLISHED_STD):
        # the published means come from an unknown (small) number of episodes: allow one published stdev + 20 %
        assert abs(m - pm) <= ps + 0.2 * pm, (m, pm, ps)
    assert 7 <= int(np.argmin(means)) <= 11            # optimum between -1 and +3 dBm (published: +1 dBm)
    # CSV shape of graph_load.py:144-186
    lines = open(names[8]).read().splitlines()
    assert lines[0].startswith("# Date:")
    assert lines[1].split(",")[:3] == ["episode", "service_blocking_rate", "episode_service_blocking_rate"]
    assert lines[1].endswith("modulation_6,episode_disrupted_services,episode_time,mean_gsnr")
    assert len(lines) == 2 + 256 and len(lines[2].split(",")) == len(lines[1].split(","))
    gsnr = np.array([float(l.split(",")[-1]) for l in lines[2:]])
    assert 14.0 < gsnr.mean() < 20.0


def test_drivers_defragmentation_columns_and_plugin_path(tmp_path):
    """graph_load-style sweep with defragmentation: the CSV's reallocation / defrag-cycle columns come from the device
    counters; a plugin-only policy runs through the reference's per-env loop and writes the same columns."""
    import csv
    sys.path.insert(0, os.path.join(REPO, "examples", "JOCN_Benchmark_2024"))
    import jocn_common as J
    from optical_networking_gym.heuristics.heuristics import heuristic_mscl_simplified
    topology = J.load_topology("nsfnet_chen.txt", 5)
    common = dict(load=250.0, num_spectrum_resources=320, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
                  launch_power_dbm=1.0, capacity=1024, defragmentation=True, n_defrag_services=0)
    names = [str(tmp_path / f"load_{ld}.csv") for ld in (200, 300)]
    res = J.run_sweep(topology, n_episodes=2, episode_length=300, replicas_per_point=2, seed=5, common=common,
                      points=[dict(load=200.0), dict(load=300.0)], monitor_names=names, policy=0)
    assert all(len(r) == 2 for r in res)
    for name in names:
        rows = list(csv.reader(l for l in open(name) if not l.startswith("#")))
        head = [h.strip() for h in rows[0]]
        assert head[5:7] == ["episode_service_realocations", "episode_defrag_cicles"]
        assert all(int(r[5]) > 0 and int(r[6]) >= int(r[5]) // 300 for r in rows[1:]) and len(rows) == 3
    common.pop("defragmentation"); common.pop("n_defrag_services")
    out = J.run_sweep_plugin(topology, heuristic_mscl_simplified, n_episodes=1, episode_length=120, seed=5, common=common,
                             points=[dict(load=300.0)], monitor_names=[str(tmp_path / "plugin.csv")])
    assert len(out) == 1 and 0.0 <= out[0][0] <= 1.0
    rows = list(csv.reader(l for l in open(tmp_path / "plugin.csv") if not l.startswith("#")))
    assert len(rows) == 2 and len(rows[1]) == len(rows[0])


@pytest.mark.parametrize("policy", [10, 11])
def test_scored_policies_drive_a_batched_sweep(tmp_path, policy):
    """graph_load.py -hi 3 (lowest fragmentation) and graph_launch_power.py -hi 5 (full MSCL) run as fused device policies
    (ids 10, 11): a small sweep writes the reference's CSV columns and sane blocking rates."""
    import csv
    import jocn_common as J
    topology = J.load_topology("nsfnet_chen.txt", 5)
    common = dict(load=300.0, num_spectrum_resources=128, bit_rate_selection="discrete", bit_rates=(10, 40, 100),
                  launch_power_dbm=1.0, capacity=512)
    names = [str(tmp_path / f"p{policy}_{ld}.csv") for ld in (150, 400)]
    res = J.run_sweep(topology, n_episodes=4, episode_length=150, replicas_per_point=4, seed=5, common=common,
                      points=[dict(load=150.0), dict(load=400.0)], monitor_names=names, policy=policy)
    assert all(len(r) == 4 and ((0.0 <= r) & (r <= 1.0)).all() for r in res)
    if policy == 10:       # sized slots + 1: a third of the decisions fail the step's own GSNR check and are rejected
        assert res[0].mean() > 0.1
    else:
        assert res[0].mean() <= res[1].mean() + 0.05
    for name in names:
        rows = list(csv.reader(l for l in open(name) if not l.startswith("#")))
        assert len(rows) == 5 and all(len(r) == len(rows[0]) for r in rows)
