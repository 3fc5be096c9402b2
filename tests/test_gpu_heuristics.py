"""Every policy of the reference's heuristics module — the version fused on device AND the one written against the plugin
API — on the device-backed compatibility env, compared, decision by decision, with what the reference's own function returned on the same network state
(tests/golden/dec_*.npz, written by make_golden.py::run_decisions from the compiled reference).

A fixture row = one env.step(): `st_action` is what was applied (the driver's decision, or a forced reject after the env
answered an action with the occupied-slots penalty, qrmsa.pyx:886-897, or raised the QoT ValueError, :925-929); `dec_<name>` holds (action, blocked_resources,
blocked_osnr) of every heuristic evaluated before that step."""
import numpy as np
import pytest

from common import jocn_modulations, load_traj, traj_requests
import optical_networking_gym.heuristics.heuristics as H
from optical_networking_gym.topology import bundled_topology_path, get_topology
from optical_networking_gym.wrappers.qrmsa_gym import QRMSAEnvWrapper

pytestmark = pytest.mark.gpu

TOPO_FILE = {"nsfnet": "nsfnet_chen.txt", "ring4": "ring_4.txt", "cost239": "cost239.txt", "nobel-eu": "nobel-eu.txt"}


def policy(name):
    """Every implementation of a policy: the public name (fused on device where one exists) and the plugin-API body."""
    if name == "psr_c":
        return [lambda env: H.heuristic_psr(env, variant="C"), lambda env: H.heuristic_psr_plugin(env, variant="C")]
    if name == "psr_o":
        return [lambda env: H.heuristic_psr(env, variant="O", coef_dist=0.7, coef_slots=1.3),
                lambda env: H.heuristic_psr_plugin(env, variant="O", coef_dist=0.7, coef_slots=1.3)]
    fns = [getattr(H, name)]
    if hasattr(H, name + "_plugin"):
        fns.append(getattr(H, name + "_plugin"))
    return fns


def replay(tag):
    meta, d = load_traj(tag)
    k = meta["k_paths"]
    topology = get_topology(bundled_topology_path(TOPO_FILE[meta["topology"]]), None, jocn_modulations(), 80, 0.2, 4.5, k)
    env = QRMSAEnvWrapper(
        topology=topology, seed=10, allow_rejection=True, load=meta["load"], episode_length=meta["episode_length"],
        num_spectrum_resources=meta["S"], launch_power_dbm=meta["launch_power_dbm"], bandwidth=meta["S"] * 12.5e9,
        frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9, bit_rate_selection="discrete",
        bit_rates=tuple(meta["bit_rates"]), margin=meta["margin"], file_name="", measure_disruptions=False, k_paths=k,
        modulations_to_consider=6, defragmentation=False, n_defrag_services=0, gen_observation=False,
        requests=traj_requests(d))
    env.reset()
    names = [meta["driver"]] + list(meta["observers"])
    fns = {n: policy(n) for n in names}
    sim = H.get_qrmsa_env(env)
    mismatches = {n: [] for n in names}
    rows = len(d["st_action"])
    first = rows - len(d["dec_" + meta["driver"]])          # rows before it are the first-fit warm-up
    assert first >= meta["warm"]
    for row in range(rows):
        cur = sim.current_service
        if row >= first:
            for n in names:
                want = d["dec_" + n][row - first]
                for which, fn in enumerate(fns[n]):
                    got = fn(env)
                    if (int(got[0]), int(bool(got[1])), int(bool(got[2]))) != tuple(int(x) for x in want):
                        mismatches[n].append((row, which, got, want.tolist()))
        if d["st_retry"][row] == 2:                 # the reference raised the QoT ValueError (qrmsa.pyx:925-929)
            with pytest.raises(ValueError, match="is not enough for service"):
                env.step(int(d["st_action"][row]))
            assert sim.current_service is cur
            continue
        _, reward, done, _, info = env.step(int(d["st_action"][row]))
        assert reward == d["st_reward"][row], row
        assert (sim.current_service is cur) == bool(d["st_retry"][row]), row
        if not d["st_retry"][row]:
            assert int(sim.topology.graph["services"][-1].accepted) == d["st_accepted"][row], row
    for n in names:
        assert not mismatches[n], (n, len(mismatches[n]), mismatches[n][:3])
    return meta, d


@pytest.mark.parametrize("tag", ["dec_nsfnet320_a", "dec_nsfnet320_b", "dec_nsfnet320_c", "dec_cost239_d"])
def test_cheap_policies_decide_like_the_reference(tag):
    meta, d = replay(tag)
    driver = d["dec_" + meta["driver"]]
    assert (driver[:, 0] == meta["reject_action"]).any()          # the fixture does exercise blocking


def test_exact_fit_fixture_exercises_the_occupied_slots_penalty():
    meta, d = load_traj("dec_nsfnet320_c")
    assert (d["st_retry"] == 1).sum() > 0 and d["st_forced"].sum() == (d["st_retry"] > 0).sum()


def test_lowest_fragmentation_decides_like_the_reference():
    replay("dec_nsfnet96_lf")


def test_mscl_decides_like_the_reference():
    replay("dec_nsfnet64_mscl")


def test_from_mask_validator_agrees_with_the_device_mask():
    meta, d = load_traj("obs_nsfnet320")
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    env = QRMSAEnvWrapper(topology=topology, seed=10, allow_rejection=True, load=meta["load"],
                          episode_length=meta["episode_length"], num_spectrum_resources=320, launch_power_dbm=0.0,
                          bandwidth=4e12, frequency_start=3e8 / 1565e-9, frequency_slot_bandwidth=12.5e9,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), margin=0, file_name="",
                          k_paths=5, modulations_to_consider=6, gen_observation=True, requests=traj_requests(d))
    obs, info = env.reset()
    for i in range(12):
        a = H.heuristic_from_mask(env, info["mask"])
        assert 0 <= a < len(info["mask"])
        obs, _, _, _, info = env.step(int(d["action"][i]))
    bad = info["mask"].copy()
    bad[int(np.flatnonzero(bad[:-1])[0])] = 0
    with pytest.raises(AssertionError):
        H.heuristic_from_mask(env, bad)
