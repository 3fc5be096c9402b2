"""Pins the CPU restatement (oracle/) against vectors captured from the compiled reference (tests/golden/)."""
import json
import os

import numpy as np
import pytest

from common import ALL_TRAJ, GOLDEN, golden_tables, holder_for, jocn_modulations, load_traj, traj_requests
from optical_networking_gym._native import ConfigHolder, F_BLOCKED_OSNR, F_BLOCKED_RESOURCES
from oracle_lib import OracleEnv

GSNR_RTOL = 1e-12   # same fp64 formula, span-by-span like the reference; libm vs numpy exp differ by ulps


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(GOLDEN, "kats_nsfnet320.json")))


@pytest.fixture(scope="module")
def nsf_env():
    h = ConfigHolder(golden_tables("nsfnet"), modulations=jocn_modulations(), num_spectrum_resources=320,
                     load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400))
    return OracleEnv(h)


def test_candidates(kats, nsf_env):
    for case in kats["candidates"]:
        row = np.unpackbits(np.array(case["row"], np.uint8), bitorder="little")[:case["S"]].astype(np.int32)
        assert nsf_env.candidates(row, case["n"]) == case["starts"]


def test_number_slots(kats, nsf_env):
    for case in kats["number_slots"]:
        assert [nsf_env.number_slots(case["bit_rate"], m) for m in range(6)] == case["slots"]


def test_codec(kats, nsf_env):
    assert nsf_env.reject_action == kats["constants"]["reject_action"]
    for case in kats["codec"]:
        if "decoded" in case:
            assert nsf_env.decode(case["action"]) == case["decoded"]
        else:
            assert nsf_env.encode(*case["encode"]) == case["action"]


def test_gn_empty_network(kats, nsf_env):
    nsf_env.set_trace(traj_requests(load_traj("traj_nsfnet320")[1])[:1])
    nsf_env.reset()
    for case in kats["gn_empty"]:
        out = nsf_env.gn(case["path_id"], case["slot"], case["n"])
        np.testing.assert_allclose(out, case["out"], rtol=GSNR_RTOL)


@pytest.mark.parametrize("tag", ALL_TRAJ + ["traj_nsfnet320_scripted"])
def test_gn_loaded_network(tag):
    meta, d = load_traj(tag)
    env = OracleEnv(holder_for(meta))
    off = d["gn_linkoff"]
    cnt_prefix = np.concatenate([[0], np.cumsum(d["gn_cnt"])])
    for i in range(len(d["gn_path"])):
        lo, hi = off[i], off[i + 1]
        counts = d["gn_cnt"][lo:hi]
        intf = d["gn_intf"][cnt_prefix[lo]:cnt_prefix[hi]]
        out = env.gn_lists(int(d["gn_path"][i]), int(d["gn_slot"][i]), int(d["gn_n"][i]), counts, intf)
        np.testing.assert_allclose(out, d["gn_out"][i], rtol=GSNR_RTOL)


def replay(tag, policy=True, **over):
    meta, d = load_traj(tag)
    env = OracleEnv(holder_for(meta, **over))
    env.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):   # constructor reset, explicit reset, episode reset (graph_load.py)
        env.reset()
    n = meta["n_steps"]
    snaps = {int(s): i for i, s in enumerate(d["snap_step"])} if "snap_step" in d else {}
    ep = 0
    for i in range(n):
        act, bres, bosnr = env.policy_first_fit()
        if policy:
            assert act == d["st_action"][i], f"step {i}: action {act} != {d['st_action'][i]}"
            assert bres == bool(d["st_bres"][i]) and bosnr == bool(d["st_bosnr"][i]), f"step {i} flags"
        else:
            act = int(d["st_action"][i])
        rc, r = env.step(act)
        assert rc == 0
        assert r["retry"] == d["st_retry"][i], f"step {i} retry"
        assert r["accepted"] == d["st_accepted"][i], f"step {i} accepted"
        assert r["reward"] == d["st_reward"][i], f"step {i} reward {r['reward']} {d['st_reward'][i]}"
        assert r["active"] == d["st_active"][i], f"step {i} active"
        if not r["retry"]:
            assert r["route"] == d["st_route"][i] and r["slot"] == d["st_slot"][i]
            assert r["terminated"] == d["st_term"][i]
            if r["accepted"]:
                assert r["modulation"] == d["st_mod"][i] and r["nslots"] == d["st_n"][i]
                keep = 3 if np.isfinite(d["st_ase"][i]) else 1      # defrag fixtures hold the provisioning-time GSNR only
                np.testing.assert_allclose([r["osnr"], r["ase"], r["nli"]][:keep],
                                           [d["st_osnr"][i], d["st_ase"][i], d["st_nli"][i]][:keep], rtol=GSNR_RTOL)
            assert env.stats()["episode_services_accepted"] == d["st_ep_acc"][i] or r["terminated"]
            if "st_dcyc" in d:   # info["episode_defrag_cicles"], info["episode_service_realocations"] (qrmsa.pyx:1008-1009)
                st = env.stats()
                assert st["step_defrag_cycles"] == d["st_dcyc"][i], f"step {i} defrag cycles"
                assert st["step_service_reallocations"] == d["st_drea"][i], f"step {i} reallocations"
        if i in snaps:
            grid = np.unpackbits(d["snap_grid"][snaps[i]], axis=1, bitorder="little")[:, :meta["S"]]
            np.testing.assert_array_equal(env.grid(), grid.astype(np.int32))
        if r["terminated"]:
            ti = meta["terminal_infos"][ep]
            s = env.stats()
            assert s["last_episode_accepted"] == ti["episode_services_accepted"]
            assert s["last_rejected"] == ti["rejected"]
            for k_, f_ in (("service_blocking_rate", "last_service_blocking_rate"),
                           ("episode_service_blocking_rate", "last_episode_service_blocking_rate"),
                           ("bit_rate_blocking_rate", "last_bit_rate_blocking_rate"),
                           ("episode_bit_rate_blocking_rate", "last_episode_bit_rate_blocking_rate")):
                assert s[f_] == pytest.approx(ti[k_], rel=1e-12, abs=1e-15), k_
            for m, mod in enumerate(jocn_modulations()):
                assert s["last_modulation_hist"][m] == ti[f"modulation_{float(mod.spectral_efficiency)}"]
            assert s["last_mean_gsnr"] == pytest.approx(ti["mean_gsnr"], rel=1e-12)
            ep += 1
            env.reset()
    assert ep == meta["episodes"]


@pytest.mark.parametrize("tag", ALL_TRAJ)
def test_first_fit_trajectory(tag):
    replay(tag, policy=True)


def test_scripted_actions_trajectory():
    """reject actions and occupied-slot actions (quirk Q5, qrmsa.pyx:886-897) through step(action)."""
    replay("traj_nsfnet320_scripted", policy=False)


def test_request_stream_is_deterministic_and_distributed():
    meta, _ = load_traj("traj_nsfnet320")
    h = holder_for(meta)
    a, b = OracleEnv(h), OracleEnv(h)
    a.seed(42); b.seed(42)
    a.reset(); b.reset()
    ra = a.run_first_fit(400); rb = b.run_first_fit(400)
    assert ra.tobytes() == rb.tobytes()
    c = OracleEnv(h, replica=1)
    c.seed(42); c.reset()
    assert c.run_first_fit(400).tobytes() != ra.tobytes()


# ---- observation() + action mask (gen_observation=True) ------------------------------------------------------------
OBS_TAGS = ["obs_nsfnet320", "obs_nsfnet320_dense"]


def obs_setup(tag):
    meta, d = load_traj(tag)
    h = holder_for(meta, capacity=1024, modulations_to_consider=meta.get("modulations_to_consider", 6))
    reqs = traj_requests(d)
    return meta, d, h, reqs


@pytest.mark.parametrize("tag", OBS_TAGS)
def test_observation_and_mask_vs_reference(tag):
    """The reference's 368-float observation and 9601-entry action mask at every step of a first-fit run."""
    meta, d, h, reqs = obs_setup(tag)
    env = OracleEnv(h)
    env.set_trace(reqs)
    for _ in range(meta["initial_resets"]):
        env.reset()
    pl = np.ctypeslib.as_array(h.struct.path_len_norm, shape=(h.struct.n_paths,))
    step = max(1, meta["steps"] // 25)
    for i in range(meta["steps"] + 1):
        if i % step == 0 or i == meta["steps"]:
            obs, mask = env.observe(pl, h.struct.max_bit_rate)
            want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]]
            np.testing.assert_array_equal(mask, want_mask, err_msg=f"mask step {i}")
            np.testing.assert_allclose(obs, d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            rc, _ = env.step(int(d["action"][i]))
            assert rc == 0


def test_observation_with_bands_vs_reference():
    """`bands` together with gen_observation=True (the reference's own driver passes bands, graph_launch_power.py:102):
    get_number_slots divides by the C band's width in Hz, so every candidate and every running service is ONE slot wide
    (quirk Q9, qrmsa.pyx:417-425, 1198-1205), while the observation's frequencies keep coming from channel_width
    (qrmsa.pyx:606-610, 678; core/osnr.pyx:259-369).  Reference run driven by its own mask, every step compared."""
    meta, d, h, reqs = obs_setup("obs_nsfnet320_bands")
    assert meta["bands"] and h.struct.nslots_channel_width == pytest.approx(1.25e10)
    env = OracleEnv(h)
    env.set_trace(reqs)
    for _ in range(meta["initial_resets"]):
        env.reset()
    pl = np.ctypeslib.as_array(h.struct.path_len_norm, shape=(h.struct.n_paths,))
    for m in range(6):
        assert env.number_slots(400.0, m) == 1
    for i in range(meta["steps"] + 1):
        obs, mask = env.observe(pl, h.struct.max_bit_rate)
        np.testing.assert_array_equal(mask, np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]], err_msg=f"mask step {i}")
        np.testing.assert_allclose(obs, d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            a = int(d["action"][i])
            if a != h.reject_action:
                assert env.decode(a) == d["decoded"][i].tolist(), i
            rc, r = env.step(a)
            assert rc == 0 and r["accepted"] == d["accepted"][i], i
            if r["accepted"]:
                assert r["nslots"] == 1


@pytest.mark.parametrize("tag", ["obs_nsfnet320_mtc4", "obs_nsfnet160_mtc2"])
def test_modulations_to_consider_window_vs_reference(tag):
    """modulations_to_consider < len(modulations): observation() first moves max_modulation_idx
    (get_max_modulation_index, qrmsa.pyx:543-581), describes the window of formats below it (:712-717) and the action codec
    addresses that window (:801-834).  Reference run driven by its own mask (lowest / highest valid action)."""
    meta, d, h, reqs = obs_setup(tag)
    Mc = meta["modulations_to_consider"]
    assert h.struct.n_mods_consider == Mc and h.reject_action == 5 * Mc * meta["S"] == meta["n_actions"] - 1
    env = OracleEnv(h)
    env.set_trace(reqs)
    for _ in range(meta["initial_resets"]):
        env.reset()
    pl = np.ctypeslib.as_array(h.struct.path_len_norm, shape=(h.struct.n_paths,))
    assert len(set(d["max_modulation_idx"].tolist())) >= 3          # the window really moves
    for i in range(meta["steps"] + 1):
        obs, mask = env.observe(pl, h.struct.max_bit_rate)          # the reference calls observation() in reset() and step()
        assert env.max_modulation_idx == d["max_modulation_idx"][i], i
        np.testing.assert_array_equal(mask, np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]], err_msg=f"mask step {i}")
        np.testing.assert_allclose(obs, d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            a = int(d["action"][i])
            if a != h.reject_action:
                assert env.decode(a) == d["decoded"][i].tolist(), i
            rc, r = env.step(a)
            assert rc == 0 and r["accepted"] == d["accepted"][i], i


# ---- load_balancing_best_modulation (heuristics.py:547-627) -----------------------------------------------------------
POLICY_ID = {"first_fit": 0, "load_balancing": 1, "highest_snr": 2}


@pytest.mark.parametrize("tag", ["traj_nsfnet320_lb", "traj_nobeleu320_lb", "traj_nsfnet128_hsnr"])
def test_other_policy_trajectory(tag):
    meta, d = load_traj(tag)
    pid = POLICY_ID[meta["policy"]]
    assert pid > 0
    env = OracleEnv(holder_for(meta))
    env.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    for i in range(meta["n_steps"]):
        act, bres, bosnr = env.policy(pid)
        assert act == d["st_action"][i], f"step {i}: action {act} != {d['st_action'][i]}"
        assert bres == bool(d["st_bres"][i]) and bosnr == bool(d["st_bosnr"][i]), f"step {i} flags"
        rc, r = env.step(act)
        assert rc == 0 and r["accepted"] == d["st_accepted"][i] and r["active"] == d["st_active"][i]
        if r["accepted"]:
            np.testing.assert_allclose(r["osnr"], d["st_osnr"][i], rtol=GSNR_RTOL)
        if r["terminated"]:
            env.reset()


# ---- measure_disruptions (qrmsa.pyx:937-952) ---------------------------------------------------------------------------
def test_measure_disruptions_trajectory():
    meta, d = load_traj("traj_nsfnet320_disr")
    assert meta["measure_disruptions"]
    env = OracleEnv(holder_for(meta, measure_disruptions=True))
    env.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    seen = 0
    for i in range(meta["n_steps"]):
        act, bres, bosnr = env.policy_first_fit()
        assert act == d["st_action"][i], i
        rc, r = env.step(act)
        assert rc == 0 and r["accepted"] == d["st_accepted"][i]
        s = env.stats()
        # info["disrupted_services"] = float(disrupted_services) / services_accepted (qrmsa.pyx:1035-1036; the counter is
        # zeroed by reset(), the divisor never is). The episode-level ratio of :1038-1041 divides two C ints under
        # cdivision and is therefore always 0 in the reference's output.
        want = d["st_disr"][i] * s["services_accepted"]
        assert abs(s["disrupted_services"] - want) < 1e-6, (i, s["disrupted_services"], want)
        assert d["st_ep_disr"][i] == (s["episode_disrupted_services"] // max(d["st_ep_acc"][i], 1) if d["st_ep_acc"][i] > 0 else 0)
        seen = max(seen, int(s["disrupted_services"]))
        if r["terminated"]:
            assert s["last_episode_disrupted"] == s["episode_disrupted_services"]
            env.reset()
    assert seen > 5


# ---- defragmentation (qrmsa.pyx:1117-1119, 1545-1639) -----------------------------------------------------------------
@pytest.mark.parametrize("tag", ["traj_nsfnet320_defrag", "traj_nsfnet320_defrag4"])
def test_defragmentation_trajectory(tag):
    """first-fit decisions, grid snapshots, per-step defrag counters and the episode's mean GSNR (which sees the OSNR
    that defragment() rewrites on moved services) against the reference run with defragmentation=True."""
    meta, d = load_traj(tag)
    assert meta["defragmentation"] and d["st_drea"].max() > 20
    replay(tag, policy=True, defragmentation=True, n_defrag_services=meta["n_defrag_services"])


# ---- the remaining policies (heuristics.py) against decisions captured from the reference -----------------------------
ORACLE_POLICY = {"shortest_available_path_lowest_spectrum_best_modulation": 3, "heuristic_load_balancing_first_fit": 4,
                 "best_modulation_load_balancing": 5, "heuristic_mscl_simplified": 6,
                 "heuristic_mscl_sequential_simplified": 7, "psr_c": 8, "heuristic_exact_fit": 9,
                 "heuristic_lowest_fragmentation": 10, "heuristic_mscl": 11}


@pytest.mark.parametrize("tag", ["dec_nsfnet320_a", "dec_nsfnet320_b", "dec_nsfnet320_c", "dec_cost239_d",
                                 "dec_nsfnet96_lf", "dec_nsfnet64_mscl"])
def test_remaining_policies_decide_like_the_reference(tag):
    """tests/golden/dec_*.npz: at every step the reference's own heuristic functions were evaluated on the same state;
    the oracle's restatements must return the same (action, blocked_resources, blocked_osnr)."""
    meta, d = load_traj(tag)
    over = dict(meta, mean_holding=10800.0, frequency_start=3e8 / 1565e-9, slot_bw=12.5e9)
    env = OracleEnv(holder_for(over))
    env.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    names = [n for n in [meta["driver"]] + list(meta["observers"]) if n in ORACLE_POLICY]
    rows = len(d["st_action"])
    first = rows - len(d["dec_" + meta["driver"]])
    checked = 0
    for row in range(rows):
        if row >= first:
            for n in names:
                got = env.policy(ORACLE_POLICY[n])
                want = d["dec_" + n][row - first]
                assert (got[0], int(got[1]), int(got[2])) == tuple(int(x) for x in want), (n, row)
                checked += 1
        rc, r = env.step(int(d["st_action"][row]))
        if d["st_retry"][row] == 2:
            assert rc != 0 and (r["flags"] & 4)          # ONGYM_F_QOT_ERROR: the reference raised its ValueError
        else:
            assert rc == 0 and r["retry"] == d["st_retry"][row] and r["reward"] == d["st_reward"][row], row
            if not r["retry"]:
                assert r["accepted"] == d["st_accepted"][row], row
    assert checked == len(names) * len(d["dec_" + meta["driver"]]) >= (750 if "320" in tag or "cost" in tag else 100)


def _epreset_checks(meta, d, recs, stats_at_term, reset_counters_called_at):
    n = meta["n_steps"]
    for f, g in (("action", "st_action"), ("accepted", "st_accepted"), ("slot", "st_slot"), ("route", "st_route"),
                 ("nslots", "st_n"), ("terminated", "st_term"), ("active", "st_active")):
        assert np.array_equal(recs[f][:n].astype(np.int64), d[g].astype(np.int64)), f
    np.testing.assert_allclose(recs["osnr"][:n], d["st_osnr"], rtol=1e-9)
    assert reset_counters_called_at == meta["reset_at"]
    ti = meta["terminal_info"]
    assert int(stats_at_term["last_episode_accepted"]) == ti["episode_services_accepted"]
    assert int(stats_at_term["last_rejected"]) == ti["rejected"]
    for ours, ref in (("last_service_blocking_rate", "service_blocking_rate"),
                      ("last_episode_service_blocking_rate", "episode_service_blocking_rate"),
                      ("last_bit_rate_blocking_rate", "bit_rate_blocking_rate"),
                      ("last_episode_bit_rate_blocking_rate", "episode_bit_rate_blocking_rate")):
        assert stats_at_term[ours] == pytest.approx(ti[ref], rel=1e-12), ours
    hist = [ti[f"modulation_{float(se)}"] for se in (1, 2, 3, 4, 5, 6)]
    assert list(stats_at_term["last_modulation_hist"][:6]) == hist
    assert stats_at_term["last_mean_gsnr"] == pytest.approx(ti["mean_gsnr"], rel=1e-9)


def test_counters_only_reset_vs_reference():
    """reset(options={"only_episode_counters": True}) (qrmsa.pyx:427-464): the episode counters restart, the departure heap is
    dropped (the services running at that moment stay for good: the active count never falls below them again), the
    episode then lasts episode_length steps, and mean Service.OSNR is taken over the whole services list."""
    meta, d = load_traj("traj_nsfnet320_epreset")
    env = OracleEnv(holder_for(meta, auto_reset=False))
    env.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    recs = np.concatenate([env.run_first_fit(meta["reset_at"])])
    env.reset_counters()
    assert env.stats()["episode_services_processed"] == 0 and env.stats()["active"] == d["st_active"][meta["reset_at"] - 1]
    recs = np.concatenate([recs, env.run_first_fit(meta["term_at"] - meta["reset_at"])])
    assert recs["terminated"][-1] == 1 and recs["terminated"][:-1].sum() == 0
    st = env.stats()
    assert d["st_active"][meta["reset_at"]:meta["term_at"]].min() >= d["st_active"][meta["reset_at"] - 1] - 1
    env.reset()
    recs = np.concatenate([recs, env.run_first_fit(meta["after_full"])])
    _epreset_checks(meta, d, recs, st, meta["reset_at"])
