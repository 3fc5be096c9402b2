"""The reference's Python surface (QRMSAEnvWrapper / heuristics plugin API / calculate_osnr) on one device replica,
driven exactly like examples/JOCN_Benchmark_2024/graph_load.py:129-164 and checked against the captured reference."""
import numpy as np
import pytest

from common import jocn_modulations, load_traj, traj_requests
from optical_networking_gym.core.osnr import calculate_osnr
from optical_networking_gym.heuristics.heuristics import (
    get_action_index, get_qrmsa_env, heuristic_highest_snr, heuristic_highest_snr_plugin, load_balancing_best_modulation,
    heuristic_shortest_available_path_first_fit_best_modulation,
    heuristic_shortest_available_path_first_fit_best_modulation_plugin)
from optical_networking_gym.topology import bundled_topology_path, get_topology
from optical_networking_gym.wrappers.qrmsa_gym import QRMSAEnvWrapper

pytestmark = pytest.mark.gpu

TOPO_FILE = {"nsfnet": "nsfnet_chen.txt", "ring4": "ring_4.txt", "cost239": "cost239.txt", "nobel-eu": "nobel-eu.txt"}


def jocn_env(tag, **over):
    meta, d = load_traj(tag)
    topology = get_topology(bundled_topology_path(TOPO_FILE[meta["topology"]]), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env_args = dict(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                    episode_length=meta["episode_length"], num_spectrum_resources=meta["S"],
                    launch_power_dbm=meta["launch_power_dbm"], bandwidth=meta["S"] * 12.5e9,
                    frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                    bit_rate_selection=meta["bit_rate_selection"], bit_rates=tuple(meta["bit_rates"]),
                    margin=meta["margin"], file_name="", measure_disruptions=False, k_paths=5,
                    modulations_to_consider=6, defragmentation=False, n_defrag_services=0, gen_observation=False,
                    requests=traj_requests(d))
    env_args.update(over)
    env = QRMSAEnvWrapper(**env_args)   # constructor resets once (qrmsa.pyx:414-415)
    env.reset()                         # graph_load.py:129-130
    return env, meta, d


def test_jocn_loop_ring4_full_episodes():
    """whole run_environment loop: two 299-step episodes, fused first-fit through the plugin signature."""
    env, meta, d = jocn_env("traj_ring4")
    i = 0
    for ep in range(meta["episodes"]):
        obs, info = env.reset()
        assert obs.shape == (368,) and obs.dtype == np.float32 and not obs.any()
        assert info["mask"].shape == (5 * 6 * meta["S"] + 1,) and not info["mask"].any()
        done = False
        while not done:
            action, bres, bosnr = heuristic_shortest_available_path_first_fit_best_modulation(env)
            assert (action, bres, bosnr) == (d["st_action"][i], bool(d["st_bres"][i]), bool(d["st_bosnr"][i]))
            _, reward, done, truncated, info = env.step(action)
            assert reward == d["st_reward"][i] and done == bool(d["st_term"][i]) and truncated is False
            assert info["episode_services_accepted"] == d["st_ep_acc"][i]
            assert info["chosen_path_index"] == d["st_route"][i] and info["chosen_slot"] == d["st_slot"][i]
            if d["st_accepted"][i]:
                assert info["osnr"] == pytest.approx(d["st_osnr"][i], rel=1e-9)
            i += 1
        ti = meta["terminal_infos"][ep]
        for k in ("service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                  "episode_bit_rate_blocking_rate"):
            assert info[k] == pytest.approx(ti[k], rel=1e-12, abs=1e-15), k
        for k in ("rejected", "blocked_due_to_resources", "blocked_due_to_osnr", "episode_services_accepted",
                  "modulation_1.0", "modulation_2.0", "modulation_3.0", "modulation_4.0", "modulation_5.0",
                  "modulation_6.0"):
            assert info[k] == ti[k], k
        services = env.env.topology.graph["services"]          # graph_load.py:181-185
        assert len(services) == ti["n_services"]
        assert sum(s.OSNR for s in services) / len(services) == pytest.approx(ti["mean_gsnr"], rel=1e-9)
    assert i == meta["n_steps"]


def test_plugin_api_policy_equals_fused_policy_and_reference():
    """A policy written only against the plugin API (get_available_slots/_get_candidates/calculate_osnr ...) picks the
    reference's actions; the graph views it reads follow the device state."""
    env, meta, d = jocn_env("traj_nsfnet320")
    env.reset()
    sim = get_qrmsa_env(env)
    for i in range(160):
        fused = heuristic_shortest_available_path_first_fit_best_modulation(env)
        plugin = heuristic_shortest_available_path_first_fit_best_modulation_plugin(env)
        assert fused == plugin == (d["st_action"][i], bool(d["st_bres"][i]), bool(d["st_bosnr"][i])), i
        env.step(fused[0])
    # views
    grid = sim.topology.graph["available_slots"]
    running = sim.topology.graph["running_services"]
    assert len(running) == d["st_active"][159]
    occupied = sum((s.number_slots + (1 if s.initial_slot + s.number_slots < 320 else 0)) * s.path.hops for s in running)
    assert (grid == 0).sum() == occupied
    svc = running[0]
    link = svc.path.links[0]
    assert svc in sim.topology[link.node1][link.node2]["running_services"]
    assert sim.topology[link.node1][link.node2]["index"] == link.id
    # helper API
    cur = sim.current_service
    path = sim.k_shortest_paths[cur.source, cur.destination][0]
    avail = sim.get_available_slots(path)
    rows = sim._get_spectrum_slots(0)
    np.testing.assert_array_equal(avail, np.prod(rows, axis=0))
    n = sim.get_number_slots(cur, sim.modulations[3])
    starts = sim._get_candidates(avail, n, 320)
    assert starts and all(sim.is_path_free(path, s, n) for s in starts[:3])
    busy = np.where(avail == 0)[0]
    assert not sim.is_path_free(path, int(busy[0]), n)
    assert sim.encoded_decimal_to_array(get_action_index(sim, 2, 3, 17)) == [2, 3, 17]
    blocks, lengths = sim.get_available_blocks(0, n, 3)
    assert all(avail[b:b + l].all() for b, l in zip(blocks, lengths))
    # highest SNR: fused device policy == the same policy written against the plugin API; the env accepts its action
    assert heuristic_highest_snr(env) == heuristic_highest_snr_plugin(env)
    assert load_balancing_best_modulation(env)[0] != -1
    action, _, _ = heuristic_highest_snr(env)
    _, reward, _, _, info = env.step(action)
    assert reward == 0.0 or action == sim.reject_action


def test_step_errors_like_the_reference():
    env, meta, d = jocn_env("traj_nsfnet320")
    env.reset()
    sim = get_qrmsa_env(env)
    # request #2 of the trace is ('2' -> '11', 10 G)... drive to a request whose best modulation fails QoT
    for i in range(400):
        cur = sim.current_service
        a, _, _ = heuristic_shortest_available_path_first_fit_best_modulation(env)
        route, mod, slot = sim.encoded_decimal_to_array(a) if a != sim.reject_action else (0, 5, 0)
        if a != sim.reject_action and mod < 5:
            bad = get_action_index(sim, route, mod + 1, slot)      # one modulation better than first-fit found feasible
            n_bad = sim.get_number_slots(cur, sim.modulations[mod + 1])
            path = sim.k_shortest_paths[cur.source, cur.destination][route]
            if sim.is_path_free(path, slot, n_bad):
                with pytest.raises(ValueError, match="is not enough for service"):
                    env.step(bad)
                assert sim.current_service is cur
                break
        env.step(a)
    else:
        pytest.skip("no QoT-limited request in the prefix")
    # occupied slots: penalty, same request (quirk Q5)
    path = sim.k_shortest_paths[cur.source, cur.destination][0]
    avail = sim.get_available_slots(path)
    busy = np.where(avail == 0)[0]
    if len(busy):
        _, reward, done, _, info = env.step(get_action_index(sim, 0, 0, int(busy[0])))
        assert reward < -3.0 and not done and sim.current_service is cur
        assert set(info) == {"blocked_due_to_resources", "blocked_due_to_osnr", "rejected", "mask"}
    # reject action
    _, reward, _, _, info = env.step(sim.reject_action)
    assert reward == -6.0 and info["chosen_path_index"] == -1 and sim.current_service is not cur


def test_reference_defaults_and_narrow_codec_construct():
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    # the reference's own defaults: continuous bit rates (25..100) and gen_observation=True (qrmsa.pyx:206-237)
    dflt = QRMSAEnvWrapper(topology=topology, load=300)
    obs, info = dflt.reset()
    assert obs.shape == (368,) and info["mask"].shape == (9601,) and info["mask"][-1] == 1 and info["mask"][:-1].any()
    assert 0.25 <= obs[0] <= 1.0            # bit rate / max(bit_rates) with bit_rates = (10, 40, 100)
    narrow = QRMSAEnvWrapper(topology=topology, load=300, gen_observation=False, modulations_to_consider=3)
    assert narrow.env.action_space.n == 5 * 3 * 320 + 1 and narrow.env.reject_action == 5 * 3 * 320   # qrmsa.pyx:313, 319-321
    assert narrow.env.encoded_decimal_to_array(1 * 3 * 320 + 2 * 320 + 7) == [1, 3, 7]                # window 5, 4, 3
    import networkx as nx
    with pytest.raises(KeyError, match="ksp"):
        QRMSAEnvWrapper(topology=nx.Graph(), gen_observation=False)


def test_gen_observation_through_the_wrapper():
    """gen_observation=True: reset()/step() return the reference's observation vector and info['mask']
    (what MaskablePPO consumes through QRMSAEnvWrapper.action_masks(), wrappers/qrmsa_gym.py:74-75)."""
    meta, d = load_traj("obs_nsfnet320")
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                          episode_length=meta["episode_length"], num_spectrum_resources=320, launch_power_dbm=0.0,
                          bandwidth=4e12, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), margin=0, file_name="",
                          k_paths=5, modulations_to_consider=6, gen_observation=True, requests=traj_requests(d))
    obs, info = env.reset()
    for i in range(25):
        want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:9601]
        assert obs.shape == (368,) and obs.dtype == np.float32
        np.testing.assert_allclose(obs, d["obs"][i], rtol=2e-6, atol=2e-7)
        np.testing.assert_array_equal(info["mask"], want_mask)
        np.testing.assert_array_equal(env.action_masks(), want_mask)
        assert info["mask"][int(d["action"][i])] == 1          # the first-fit action is always inside the mask
        obs, reward, done, truncated, info = env.step(int(d["action"][i]))


def test_step_bundle_prefetches_the_next_decision():
    """QRMSAEnv.step() asks the device for the next decision of the fused heuristic the caller used last in the same call
    (ongym_step_actions_bundle: one synchronisation per loop iteration of graph_load.py:157-164); the prefetched decision must
    be the one a fresh policy call gives, also when the caller switches heuristics, resets or retries."""
    from optical_networking_gym import _native as nat
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=3, load=1500, episode_length=120, num_spectrum_resources=320,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), gen_observation=False)
    sim = get_qrmsa_env(env)
    env.reset()
    hits = 0
    for i in range(400):
        pid = nat.POLICY_LOAD_BALANCING if (i // 50) % 2 else nat.POLICY_FIRST_FIT
        fresh_a, fresh_f = sim._dev.policy_actions(pid)
        cached = sim._policy_cache is not None and sim._policy_cache[0] == pid and sim._policy_cache[1] == sim._state_version
        hits += int(cached)
        a, bres, bosnr = sim.policy_action(pid)
        assert a == int(fresh_a[0]) and bres == bool(fresh_f[0] & nat.F_BLOCKED_RESOURCES) and bosnr == bool(fresh_f[0] & nat.F_BLOCKED_OSNR)
        if i % 37 == 5:            # an occupied-slots action: penalty, the same request stays (quirk Q5), nothing may be cached
            busy = np.where(sim.get_available_slots(sim.k_shortest_paths[sim.current_service.source, sim.current_service.destination][0]) == 0)[0]
            if len(busy):
                env.step(int(busy[0]))
                assert sim._policy_cache is None
        _, _, done, _, _ = env.step(a)
        if done:
            env.reset()
    assert hits > 300


def test_bands_with_gen_observation_through_the_wrapper():
    """The reference's own driver passes bands=[S, C, L] (graph_launch_power.py:102) and its observation runs with it: one slot
    per service (quirk Q9), frequencies from channel_width.  QRMSAEnvWrapper(bands=..., gen_observation=True) against the
    reference's captured run (obs_nsfnet320_bands), driven by the reference's own mask."""
    from optical_networking_gym.core.bands import BandC, BandL, BandS
    meta, d = load_traj("obs_nsfnet320_bands")
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                          episode_length=meta["episode_length"], num_spectrum_resources=320, launch_power_dbm=0.0,
                          bandwidth=4e12, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), margin=0, file_name="",
                          k_paths=5, modulations_to_consider=6, gen_observation=True, requests=traj_requests(d),
                          bands=[BandS(), BandC(), BandL()])
    sim = get_qrmsa_env(env)
    assert sim.current_band.name == "C" and sim.channel_width == 12.5
    obs, info = env.reset()
    for i in range(40):
        want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:9601]
        np.testing.assert_allclose(obs, d["obs"][i], rtol=2e-6, atol=2e-7)
        np.testing.assert_array_equal(info["mask"], want_mask)
        assert all(sim.get_number_slots(sim.current_service, m) == 1 for m in sim.modulations)
        obs, reward, done, truncated, info = env.step(int(d["action"][i]))
        if d["accepted"][i]:
            assert info["chosen_path_index"] == d["decoded"][i][0] and info["chosen_slot"] == d["decoded"][i][2]


def test_bands_quirk_and_service_csv(tmp_path):
    """bands=[S, C, L] (graph_launch_power.py:108): every service needs ONE slot (quirk Q9: SURVEY measured
    400 G -> [1,1,1,1,1,1] in the reference); file_name: per-service CSV with the reference's header (qrmsa.pyx:387-406)."""
    from optical_networking_gym.core.bands import BandC, BandL, BandS
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    prefix = str(tmp_path / "svc")
    env = QRMSAEnvWrapper(topology=topology, seed=7, load=300, episode_length=40, num_spectrum_resources=320,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), gen_observation=False,
                          bands=[BandS(), BandC(), BandL()], file_name=prefix, measure_disruptions=True)
    sim = get_qrmsa_env(env)
    assert sim.current_band.name == "C"
    env.reset()
    done = False
    while not done:
        cur = sim.current_service
        assert [sim.get_number_slots(cur, m) for m in sim.modulations] == [1, 1, 1, 1, 1, 1]
        action, _, _ = heuristic_shortest_available_path_first_fit_best_modulation(env)
        _, _, done, _, info = env.step(action)
    assert all(s.number_slots == 1 for s in sim.topology.graph["services"] if s.accepted)
    env.close()
    lines = open(sim.final_file_name).read().splitlines()
    assert lines[0] == "# Service stats file from simulator"
    assert lines[1] == ("id,source,destination,bit_rate,path_k,path_length,modulation,min_osnr,osnr,ase,nli,"
                        "disrupted_services,active_services")
    assert len(lines) == 2 + 39 and len(lines[2].split(",")) == 13
    assert sim.final_file_name.endswith("_nsfnet_chen_0.0_300.0_7.csv")


def test_defragmentation_through_the_wrapper():
    """defragmentation=True (graph_load.py passes it through env_args): info counters per step, and the episode's mean
    GSNR over topology.graph["services"], whose moved members carry the OSNR defragment() rewrote."""
    env, meta, d = jocn_env("traj_nsfnet320_defrag", defragmentation=True, n_defrag_services=0)
    sim = get_qrmsa_env(env)
    env.reset()
    i = 0
    done = False
    while not done:
        action, _, _ = heuristic_shortest_available_path_first_fit_best_modulation(env)
        assert action == d["st_action"][i]
        _, _, done, _, info = env.step(action)
        assert info["episode_defrag_cicles"] == d["st_dcyc"][i], i
        assert info["episode_service_realocations"] == d["st_drea"][i], i
        i += 1
    ti = meta["terminal_infos"][0]
    services = sim.topology.graph["services"]
    assert len(services) == ti["n_services"] and d["st_drea"][i - 1] > 20
    assert sum(s.OSNR for s in services) / len(services) == pytest.approx(ti["mean_gsnr"], rel=1e-9)


def test_link_statistics_like_the_reference():
    """_update_link_stats (qrmsa.pyx:1353-1480) on the device-backed env against values obtained by calling the
    reference's method on the same states (tests/golden/linkstats_nsfnet320.*).  _get_network_compactness (:1150-1186)
    segfaults in the compiled reference, so it has no fixture: PARITY UNPINNED - it is checked against a slot-by-slot
    recomputation of the source text's definition on the same state."""
    import json, os
    from common import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "linkstats_nsfnet320.json")))
    d = np.load(os.path.join(GOLDEN, "linkstats_nsfnet320.npz"))
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                          episode_length=meta["episode_length"], num_spectrum_resources=meta["S"], launch_power_dbm=0.0,
                          bandwidth=meta["S"] * 12.5e9, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), margin=0, file_name="", k_paths=5,
                          modulations_to_consider=6, gen_observation=False, requests=traj_requests(d))
    env.reset()
    sim = get_qrmsa_env(env)
    checks = {c["step"]: c for c in meta["checks"]}
    assert [list(map(str, e)) for e in sim.topology.edges()] == meta["edges"]
    for i, action in enumerate(d["st_action"]):
        got = heuristic_shortest_available_path_first_fit_best_modulation(env)[0]
        assert got == action
        env.step(int(action))
        if i in checks:
            assert sim.current_time == checks[i]["current_time"]
            for (u, v), want in zip(sim.topology.edges(), checks[i]["links"]):
                sim._update_link_stats(u, v)
                link = sim.topology[u][v]
                have = [link["utilization"], link["external_fragmentation"], link["compactness"], link["last_update"]]
                np.testing.assert_allclose(have, want, rtol=1e-12, atol=1e-15)
            # network compactness, recomputed slot by slot (parity unpinned, see the docstring)
            grid = np.asarray(sim.topology.graph["available_slots"])
            slot_hops = sum(s.number_slots * s.path.hops for s in sim.topology.graph["running_services"])
            occupied = unused = 0
            for u, v in sim.topology.edges():
                row = [int(x) for x in grid[sim.topology[u][v]["index"]]]
                blocks = [j for j in range(len(row)) if row[j] == 0 and (j == 0 or row[j - 1] != 0)]      # used-block starts
                if len(blocks) > 1:
                    lo = blocks[0]
                    hi = max(j for j in range(len(row)) if row[j] == 0) + 1
                    occupied += hi - lo
                    unused += sum(1 for j in range(lo, hi) if row[j] == 1 and (j == lo or row[j - 1] != 1))
            want_c = (occupied / slot_hops) * (sim.topology.number_of_edges() / unused) if unused else 1.0
            assert sim._get_network_compactness() == pytest.approx(want_c, rel=1e-12)
            assert unused > 0 and want_c > 0


def test_counters_only_reset_through_the_gym_surface():
    """env.reset(options={"only_episode_counters": True}) (qrmsa.pyx:427-464) on the compatibility env, driven like the
    reference run that produced tests/golden/traj_nsfnet320_epreset: same actions (including the one decision that only
    comes out right when running services with the new request's service_id are skipped, core/osnr.pyx:65), the episode
    then lasts episode_length steps, terminal info and mean GSNR over the un-cleared services list."""
    meta, d = load_traj("traj_nsfnet320_epreset")
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                          episode_length=meta["episode_length"], num_spectrum_resources=meta["S"], launch_power_dbm=0.0,
                          bandwidth=meta["S"] * 12.5e9, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                          bit_rate_selection="discrete", bit_rates=tuple(meta["bit_rates"]), margin=0.0, file_name="",
                          measure_disruptions=False, k_paths=5, modulations_to_consider=6, defragmentation=False,
                          n_defrag_services=0, gen_observation=False, requests=traj_requests(d))
    env.reset()
    ff = heuristic_shortest_available_path_first_fit_best_modulation
    i = 0

    def one():
        nonlocal i
        action, _, _ = ff(env)
        assert action == d["st_action"][i], i
        _, _, done, _, info = env.step(action)
        assert done == bool(d["st_term"][i]) and info["episode_services_accepted"] == d["st_ep_acc"][i], i
        assert info["episode_service_blocking_rate"] == pytest.approx(d["st_ep_blk"][i], rel=1e-12, abs=1e-15)
        assert info["episode_bit_rate_blocking_rate"] == pytest.approx(d["st_ep_brblk"][i], rel=1e-12, abs=1e-15)
        assert len(env.env.topology.graph["running_services"]) == d["st_active"][i]
        i += 1
        return done, info

    for _ in range(meta["before"]):
        one()
    cur = env.env.current_service
    obs, info = env.reset(options={"only_episode_counters": True})
    assert info == {} and not obs.any() and env.env.current_service is cur
    done = False
    while not done:
        done, info = one()
    assert i == meta["term_at"]
    ti = meta["terminal_info"]
    for k in ("rejected", "episode_services_accepted", "modulation_2.0", "modulation_3.0", "modulation_4.0", "modulation_6.0"):
        assert info[k] == ti[k], k
    for k in ("service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate", "episode_bit_rate_blocking_rate"):
        assert info[k] == pytest.approx(ti[k], rel=1e-12), k
    services = env.env.topology.graph["services"]
    assert len(services) == ti["n_services"]
    assert sum(s.OSNR for s in services) / len(services) == pytest.approx(ti["mean_gsnr"], rel=1e-9)
    env.reset()
    for _ in range(meta["after_full"]):
        one()


def test_env_k_paths_below_the_topology_k():
    """QRMSAEnv(topology built with k=5, k_paths=3): routes, action space, reject action and the device tables all use the
    first three routes (a heuristic's reject action then IS the device's; ADVICE r1: a mismatch used to loop forever or
    decode a reject as an allocation).  Decisions equal a topology built with k=3 outright."""
    mods = jocn_modulations()
    topo5 = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, mods, 80, 0.2, 4.5, 5)
    topo3 = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, mods, 80, 0.2, 4.5, 3)
    kw = dict(seed=3, allow_rejection=True, load=900, episode_length=400, num_spectrum_resources=160, launch_power_dbm=0.0,
              bandwidth=160 * 12.5e9, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), margin=0.0, file_name="", k_paths=3,
              modulations_to_consider=6, gen_observation=False)
    a, b = QRMSAEnvWrapper(topology=topo5, **kw), QRMSAEnvWrapper(topology=topo3, **kw)
    assert a.env.action_space.n == b.env.action_space.n == 3 * 6 * 160 + 1 and a.env.reject_action == 3 * 6 * 160
    assert all(len(r) == 3 for r in a.env.k_shortest_paths.values())
    a.reset(); b.reset()
    rejected = 0
    for i in range(300):
        fa = heuristic_shortest_available_path_first_fit_best_modulation(a)
        fb = heuristic_shortest_available_path_first_fit_best_modulation_plugin(b)
        assert fa == fb, i
        rejected += fa[0] == a.env.reject_action
        ra, rb = a.step(fa[0]), b.step(fb[0])
        assert ra[1] == rb[1] and ra[2] == rb[2] and ra[4]["episode_services_accepted"] == rb[4]["episode_services_accepted"]
    assert rejected > 5          # the reject action really went through step() (reward -6, next request drawn)
    with pytest.raises(ValueError):
        QRMSAEnvWrapper(topology=topo3, **dict(kw, k_paths=5))
