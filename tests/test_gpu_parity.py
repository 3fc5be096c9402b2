"""Parity of the HIP path (through the C ABI) against (a) the golden vectors captured from the reference and
(b) the CPU oracle on seeded random traffic.  Bar: bit-exact for actions / slots / blocking decisions / counters,
GSNR within GSNR_RTOL (the north star allows 1e-5 relative; the fp64 span-hoisted device sum is held to 1e-9)."""
import numpy as np
import pytest

from common import ALL_TRAJ, golden_tables, holder_for, jocn_modulations, load_traj, traj_requests
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv, OngymError
from oracle_lib import OracleEnv

pytestmark = pytest.mark.gpu
LEAN_POLICIES = (nat.POLICY_FIRST_FIT, nat.POLICY_LOAD_BALANCING, nat.POLICY_HIGHEST_SNR, nat.POLICY_LOWEST_FRAGMENTATION)

GSNR_RTOL = 1e-9
EXACT = ("action", "route", "modulation", "slot", "nslots", "accepted", "terminated", "retry", "flags", "active",
         "reward")


def make_env(meta, batch=1, **over):
    kw = dict(tables=golden_tables(meta["topology"]), modulations=jocn_modulations(), batch_size=batch,
              num_spectrum_resources=meta["S"], episode_length=meta["episode_length"], load=meta["load"],
              mean_service_holding_time=meta["mean_holding"], bit_rate_selection=meta["bit_rate_selection"],
              bit_rates=tuple(meta["bit_rates"]), bit_rate_lower_bound=25, bit_rate_higher_bound=100,
              launch_power_dbm=meta["launch_power_dbm"], frequency_start=meta["frequency_start"],
              frequency_slot_bandwidth=meta["slot_bw"], margin=meta["margin"], capacity=1024,
              nslots_channel_width=meta.get("nslots_channel_width", 0.0))
    kw.update(over)
    return BatchedQRMSAEnv(**kw)


def assert_records_equal(got, want, ctx=""):
    for f in EXACT:
        if not np.array_equal(got[f], want[f]):
            bad = np.argwhere(got[f] != want[f])[0]
            raise AssertionError(f"{ctx}: field {f} differs first at {tuple(bad)}: {got[f][tuple(bad)]} != {want[f][tuple(bad)]}")
    for f in ("osnr", "ase", "nli"):
        np.testing.assert_allclose(got[f], want[f], rtol=GSNR_RTOL, err_msg=f"{ctx}: {f}")


@pytest.mark.parametrize("tag,generic", [(t, g) for t in ALL_TRAJ for g in (False, True)])
def test_first_fit_trajectory_vs_reference(tag, generic, monkeypatch):
    """Replays the reference's captured request trace; every step of the fused policy+step kernel must reproduce the
    reference's action, slot, modulation, accept decision, reward, termination and GSNR.  Both kernels: the lean k_fast
    (the benchmark's; it replays traces whose bit rates come from the configured table) and the generic k_run
    (ONGYM_FORCE_GENERIC=1)."""
    meta, d = load_traj(tag)
    if generic:
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
    env = make_env(meta, auto_reset=True)
    env.set_requests(traj_requests(d))
    lean_expected = (not generic) and meta["bit_rate_selection"] == "discrete"
    assert env.occupancy()["lean_kernel"] == lean_expected
    for _ in range(meta["initial_resets"]):
        env.reset()
    n = meta["n_steps"]
    rec = env.step_policy(n)[:, 0]
    assert np.array_equal(rec["action"], d["st_action"])
    assert np.array_equal(rec["accepted"], d["st_accepted"])
    assert np.array_equal(rec["terminated"], d["st_term"])
    assert np.array_equal(rec["reward"], d["st_reward"])
    assert np.array_equal(rec["active"], d["st_active"])
    assert np.array_equal(rec["route"], d["st_route"])
    assert np.array_equal(rec["slot"], d["st_slot"])
    acc = d["st_accepted"] == 1
    assert np.array_equal(rec["modulation"][acc], d["st_mod"][acc])
    assert np.array_equal(rec["nslots"][acc], d["st_n"][acc])
    assert np.array_equal((rec["flags"] & nat.F_BLOCKED_RESOURCES) != 0, d["st_bres"] == 1)
    assert np.array_equal((rec["flags"] & nat.F_BLOCKED_OSNR) != 0, d["st_bosnr"] == 1)
    for f, g in (("osnr", "st_osnr"), ("ase", "st_ase"), ("nli", "st_nli")):
        np.testing.assert_allclose(rec[f], d[g], rtol=GSNR_RTOL)
    st = env.stats()[0]
    ti = meta["terminal_infos"][-1]
    assert st["episodes_completed"] == meta["episodes"]
    assert st["last_episode_accepted"] == ti["episode_services_accepted"]
    assert st["last_rejected"] == ti["rejected"]
    assert st["last_service_blocking_rate"] == pytest.approx(ti["service_blocking_rate"], rel=1e-12, abs=1e-15)
    assert st["last_episode_bit_rate_blocking_rate"] == pytest.approx(ti["episode_bit_rate_blocking_rate"], rel=1e-12, abs=1e-15)
    assert st["last_mean_gsnr"] == pytest.approx(ti["mean_gsnr"], rel=1e-9)
    for m, mod in enumerate(jocn_modulations()):
        assert st["last_modulation_hist"][m] == ti[f"modulation_{float(mod.spectral_efficiency)}"]


def test_grid_snapshots_vs_reference():
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta, auto_reset=False)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    done = 0
    for k, s in enumerate(d["snap_step"]):
        env.step_policy(int(s) + 1 - done, record=False)
        done = int(s) + 1
        want = np.unpackbits(d["snap_grid"][k], axis=1, bitorder="little")[:, :meta["S"]].astype(np.int32)
        np.testing.assert_array_equal(env.grid(0), want)


def test_scripted_actions_vs_reference():
    """step(action) with reject actions and occupied-slot actions (quirk Q5), one launch per step."""
    meta, d = load_traj("traj_nsfnet320_scripted")
    env = make_env(meta, auto_reset=False)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    for i in range(meta["n_steps"]):
        r = env.step(np.array([d["st_action"][i]], np.int32))[0]
        assert r["retry"] == d["st_retry"][i], i
        assert r["accepted"] == d["st_accepted"][i], i
        assert r["reward"] == d["st_reward"][i], i
        assert r["active"] == d["st_active"][i], i
        if not r["retry"]:
            assert r["terminated"] == d["st_term"][i]
            assert r["route"] == d["st_route"][i] and r["slot"] == d["st_slot"][i]
            if r["accepted"]:
                assert r["modulation"] == d["st_mod"][i] and r["nslots"] == d["st_n"][i]
                np.testing.assert_allclose(r["osnr"], d["st_osnr"][i], rtol=GSNR_RTOL)


def test_gn_known_answers_vs_reference():
    """calculate_osnr on the empty network: every modulation / several paths and slots (core/osnr.pyx:21-142)."""
    import json, os
    from common import GOLDEN
    kats = json.load(open(os.path.join(GOLDEN, "kats_nsfnet320.json")))
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta)
    env.set_requests(traj_requests(d)[:2])
    env.reset()
    for case in kats["gn_empty"][::7]:
        out = env.gsnr(0, case["path_id"], case["slot"], case["n"])
        np.testing.assert_allclose(out, case["out"], rtol=GSNR_RTOL)


def run_oracle_batch(holder, seed, nsteps, batch):
    recs = np.zeros((nsteps, batch), nat.STEP_DTYPE)
    envs = []
    for r in range(batch):
        o = OracleEnv(holder, replica=r)
        o.seed(seed)
        o.reset()
        recs[:, r] = o.run_first_fit(nsteps)
        envs.append(o)
    return recs, envs


@pytest.mark.parametrize("topo,S,load,steps", [("nsfnet", 320, 300, 1300), ("cost239", 320, 400, 700),
                                                ("nobel-eu", 768, 600, 700), ("germany50", 320, 500, 400),
                                                # slot counts that are not a multiple of the bitmap word (lean kernel: 32)
                                                ("nsfnet", 100, 120, 500), ("nsfnet", 333, 350, 500), ("nobel-eu", 417, 500, 400)])
def test_random_traffic_vs_oracle(topo, S, load, steps):
    """Device request generator + fused step vs the CPU oracle on the same (seed, replica) streams, B=48 replicas with
    per-replica load / launch power / margin overrides; crosses an episode boundary (auto-reset)."""
    B = 48
    rng = np.random.default_rng(1)
    loads = load * rng.uniform(0.5, 1.6, B)
    lps = rng.uniform(-4.0, 3.0, B)
    margins = rng.choice([0.0, 0.5, 1.0], B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=1024, episode_length=1000,
              auto_reset=True, load=load, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
              replica_load=loads, replica_launch_power_dbm=lps, replica_margin=margins)
    holder = nat.ConfigHolder(golden_tables(topo), **kw)
    want, oracles = run_oracle_batch(holder, 2024, steps, B)
    env = BatchedQRMSAEnv(tables=golden_tables(topo), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=S, capacity=1024, episode_length=1000, auto_reset=True, load=load,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), replica_load=loads,
                          replica_launch_power_dbm=lps, replica_margin=margins)
    env.seed(2024)
    env.reset()
    got = env.step_policy(steps)
    assert_records_equal(got, want, f"{topo}")
    st = env.stats()
    for r in (0, 7, B - 1):
        o = oracles[r].stats()
        for f in ("services_processed", "services_accepted", "episode_services_processed", "episode_services_accepted",
                  "bit_rate_requested", "bit_rate_provisioned", "episode_bit_rate_provisioned", "rejected",
                  "episodes_completed", "total_steps", "total_accepted", "total_paths_tried", "total_path_hops",
                  "total_active_sum",
                  "current_time", "active", "last_episode_accepted", "last_service_blocking_rate"):
            assert st[r][f] == o[f], (r, f, st[r][f], o[f])
        # the device may settle an evaluation by the ASE-only bound instead of the full interferer sum
        # (k_run: per candidate, so the sum is exact; the lean k_fast: per modulation before the bitmap is read, so
        # it also counts modulations that had no candidate)
        assert st[r]["total_gn_evals"] <= o["total_gn_evals"] <= st[r]["total_gn_evals"] + st[r]["total_gn_shortcuts"]
        assert st[r]["total_interferer_terms"] <= o["total_interferer_terms"]
        np.testing.assert_array_equal(st[r]["episode_modulation_hist"], o["episode_modulation_hist"])
        np.testing.assert_array_equal(env.grid(r), oracles[r].grid())
        # plugin-API queries on a loaded network
        q = env.request(r)
        assert q.tobytes() == oracles[r].request().tobytes()
        tb = golden_tables(topo)
        p = int(tb.pair_paths[q["source"], q["destination"], 0])
        np.testing.assert_array_equal(env.available_slots(r, p), oracles[r].available(p))
        starts = oracles[r].candidates(oracles[r].available(p), 4)
        for s0 in starts[:1] + starts[-1:]:
            np.testing.assert_allclose(env.gsnr(r, p, s0, 4), oracles[r].gn(p, s0, 4), rtol=GSNR_RTOL)
        a = np.sort(env.services(r), order=["release_time", "path_id", "slot"])
        b = np.sort(oracles[r].services(), order=["release_time", "path_id", "slot"])
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("generic", [False, True], ids=["lean", "generic"])
@pytest.mark.parametrize("case", range(14))
def test_randomised_configurations_vs_oracle(case, generic, monkeypatch):
    """Configurations drawn at random (topology, slot count - mostly not a multiple of the bitmap word -, routes per pair, bit
    rate set, load, launch power, margin, episode length): fused first fit + step on device vs the
    oracle, records bit-exact, crossing several episode boundaries."""
    if generic:
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")       # k_run instead of k_fast (read at ongym_create)
    rng = np.random.default_rng(1000 + case)
    topo = ["nsfnet", "cost239", "ring4", "nobel-eu"][int(rng.integers(0, 4))]
    tb = golden_tables(topo)
    k = int(rng.integers(1, tb.k_paths + 1))
    if k < tb.k_paths:
        tb = tb.truncated(k)
    S = int(rng.integers(40, 420))
    all_rates = np.array([10, 25, 40, 100, 200, 400])
    rates = tuple(int(x) for x in np.sort(rng.choice(all_rates, size=int(rng.integers(1, 5)), replace=False)))
    B, steps = 6, 950
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(80, 250) * S / 100),
              bit_rate_selection="discrete", bit_rates=rates, auto_reset=True, episode_length=int(rng.integers(200, 450)),
              launch_power_dbm=float(rng.uniform(-3, 3)), margin=float(rng.choice([0.0, 0.5, 1.0])))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(77 + case); env.reset()
    assert env.occupancy()["lean_kernel"] == (not generic)       # every drawn configuration is eligible for k_fast
    got = env.step_policy(steps)
    rejected = 0
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(77 + case); o.reset()
        want = o.run_first_fit(steps)
        assert_records_equal(got[:, r], want, f"case {case}: {topo} S={S} k={k} rates={rates} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())
        rejected += int((want["accepted"] == 0).sum())
    assert got["terminated"].sum() >= 2 * B


@pytest.mark.parametrize("pid", list(range(1, 12)))
@pytest.mark.parametrize("case", range(3))
def test_policies_randomised_configurations_vs_oracle(pid, case, monkeypatch):
    """Every fused policy other than first fit (ids 1..11) on two randomly drawn configurations each, after a first-fit
    warm-up: step records and grids equal to the oracle's."""
    rng = np.random.default_rng(7000 + 31 * pid + case)
    topo = ["nsfnet", "cost239", "ring4"][int(rng.integers(0, 3))]
    if case == 2:
        topo = "nobel-eu"            # 41 links: the generic record codec (no link mask in the record)
    tb = golden_tables(topo)
    k = int(rng.integers(2, tb.k_paths + 1))
    if k < tb.k_paths:
        tb = tb.truncated(k)
    S = int(rng.integers(48, (100 if topo == "nobel-eu" else 140) if pid >= 10 else 300))
    rates = tuple(int(x) for x in np.sort(rng.choice(np.array([10, 40, 100, 200, 400]), size=int(rng.integers(2, 5)), replace=False)))
    B, warm, steps = 4, int(rng.integers(100, 300)), 90 if pid >= 10 else 260
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(90, 220) * S / 100),
              bit_rate_selection="discrete", bit_rates=rates, auto_reset=True, episode_length=1000,
              launch_power_dbm=float(rng.uniform(-2, 3)), margin=float(rng.choice([0.0, 0.5])))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    want = []
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(3 + case); o.reset(); o.run_policy(0, warm)
        want.append(o.run_policy(pid, steps))
    for generic in ((False, True) if pid in LEAN_POLICIES else (True,)):     # both kernels where a lean one exists
        if generic:
            monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
        env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
        env.seed(3 + case); env.reset()
        assert env.occupancy(pid)["lean_kernel"] == (not generic)
        env.step_policy(warm, record=False)
        got = env.step_policy(steps, policy=pid)
        for r in range(B):
            assert_records_equal(got[:, r], want[r], f"policy {pid} case {case} generic={generic}: {topo} S={S} k={k} rates={rates} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("case", range(6))
def test_random_external_actions_vs_oracle(case):
    """env.step(actions) with a mix of the first-fit action, the reject action and actions drawn uniformly from the whole
    codec (most decode to occupied slots: the penalty of quirk Q5, some to a QoT failure: flagged, nothing applied, the
    oracle's step returns its error code there) on random configurations; every record, then grids and services."""
    rng = np.random.default_rng(4000 + case)
    topo = ["nsfnet", "cost239", "nobel-eu"][int(rng.integers(0, 3))]
    tb = golden_tables(topo)
    S = int(rng.integers(48, 200))
    rates = tuple(int(x) for x in np.sort(rng.choice(np.array([10, 40, 100, 200, 400]), size=3, replace=False)))
    B, warm, steps = 5, 200, 220
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(120, 260) * S / 100),
              bit_rate_selection="discrete", bit_rates=rates, auto_reset=False, episode_length=10 ** 6,
              launch_power_dbm=float(rng.uniform(-2, 2)), margin=0.0)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(5 + case); env.reset()
    env.step_policy(warm, record=False)
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(5 + case); o.reset(); o.run_first_fit(warm)
        oracles.append(o)
    kinds = np.zeros(4, int)
    for i in range(steps):
        ff, _ = env.policy_actions()
        u = rng.random(B)
        acts = np.where(u < 0.5, ff, np.where(u < 0.65, env.reject_action, rng.integers(0, env.reject_action + 1, B))).astype(np.int32)
        rec = env.step(acts)
        for r, o in enumerate(oracles):
            rc, want = o.step(int(acts[r]))
            got = rec[r]
            if rc != 0:                                           # the reference raises its QoT ValueError here
                assert got["flags"] & nat.F_QOT_ERROR and not got["accepted"], (i, r)
                kinds[3] += 1
                continue
            assert not (got["flags"] & nat.F_QOT_ERROR), (i, r)
            for f in ("action", "accepted", "retry", "route", "slot", "modulation", "nslots", "reward", "terminated", "active"):
                assert got[f] == want[f], (i, r, f, got[f], want[f])
            if got["accepted"]:
                np.testing.assert_allclose(got["osnr"], want["osnr"], rtol=GSNR_RTOL)
            kinds[0 if got["accepted"] else (1 if got["retry"] else 2)] += 1
    for r, o in enumerate(oracles):
        np.testing.assert_array_equal(env.grid(r), o.grid())
        a = np.sort(env.services(r), order=["release_time", "path_id", "slot"])
        b = np.sort(o.services(), order=["release_time", "path_id", "slot"])
        assert a.tobytes() == b.tobytes()
    assert kinds[0] > 100 and kinds[1] > 50 and kinds[2] > 50, kinds      # accepted, occupied-slot retries, rejections


@pytest.mark.parametrize("feature", ["defragmentation", "measure_disruptions", "track_service_ids"])
@pytest.mark.parametrize("case", range(3))
def test_stateful_features_randomised_configurations_vs_oracle(feature, case):
    """defragmentation (random n_defrag_services), measure_disruptions and service-id tracking with a counters-only reset in
    the middle, each on three randomly drawn configurations: records, counters, grids and running services vs the oracle."""
    rng = np.random.default_rng(9000 + 17 * case + len(feature))
    topo = ["nsfnet", "cost239", "nobel-eu"][int(rng.integers(0, 3))]
    tb = golden_tables(topo)
    S = int(rng.integers(64, 260))
    rates = tuple(int(x) for x in np.sort(rng.choice(np.array([10, 40, 100, 200, 400]), size=3, replace=False)))
    B, steps = 5, 520
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(110, 230) * S / 100),
              bit_rate_selection="discrete", bit_rates=rates, auto_reset=True, episode_length=int(rng.integers(180, 400)),
              launch_power_dbm=float(rng.uniform(-2, 3)), margin=float(rng.choice([0.0, 0.5])))
    if feature == "defragmentation":
        kw.update(defragmentation=True, n_defrag_services=int(rng.integers(0, 8)))
    elif feature == "measure_disruptions":
        kw.update(measure_disruptions=True)
    else:
        kw.update(track_service_ids=True)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(21 + case); env.reset()
    half = steps // 2
    got1 = env.step_policy(half)
    if feature == "track_service_ids":
        env.reset_episode_counters()
    got2 = env.step_policy(steps - half)
    st = env.stats()
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(21 + case); o.reset()
        want1 = o.run_first_fit(half)
        if feature == "track_service_ids":
            o.reset_counters()
        want2 = o.run_first_fit(steps - half)
        assert_records_equal(got1[:, r], want1, f"{feature} case {case} {topo} S={S} replica {r} (first half)")
        assert_records_equal(got2[:, r], want2, f"{feature} case {case} {topo} S={S} replica {r} (second half)")
        so = o.stats()
        for f in ("services_accepted", "episodes_completed", "active", "disrupted_services", "episode_disrupted_services",
                  "episode_defrag_cycles", "episode_service_reallocations", "episode_services_processed", "rejected"):
            assert st[r][f] == so[f], (feature, case, r, f, st[r][f], so[f])
        np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("kind", ["continuous", "per_link_alpha", "germany50"])
@pytest.mark.parametrize("case", range(2))
def test_generic_kernel_configurations_randomised_vs_oracle(kind, case):
    """What only the generic k_run serves - continuous bit rates (random bounds), per-link attenuation, a topology with more
    than 52 links - on random configurations: fused first fit + step vs the oracle."""
    import copy
    rng = np.random.default_rng(12000 + 7 * case + len(kind))
    topo = "germany50" if kind == "germany50" else ["nsfnet", "cost239", "nobel-eu"][int(rng.integers(0, 3))]
    tb = golden_tables(topo)
    S = int(rng.integers(64, 300))
    B, steps = 5, 600
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(100, 220) * S / 100),
              auto_reset=True, episode_length=int(rng.integers(200, 400)), launch_power_dbm=float(rng.uniform(-2, 3)),
              margin=float(rng.choice([0.0, 0.5])))
    if kind == "continuous":
        lo = int(rng.integers(10, 120))
        kw.update(bit_rate_selection="continuous", bit_rates=(10, 40, 100), bit_rate_lower_bound=lo,
                  bit_rate_higher_bound=lo + int(rng.integers(5, 300)))
    else:
        kw.update(bit_rate_selection="discrete",
                  bit_rates=tuple(int(x) for x in np.sort(rng.choice(np.array([10, 40, 100, 200, 400]), size=3, replace=False))))
    if kind == "per_link_alpha":
        tb = copy.deepcopy(tb)
        tb.link_alpha = tb.link_alpha * rng.uniform(0.85, 1.25, tb.n_links)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(40 + case); env.reset()
    assert not env.occupancy()["lean_kernel"]
    got = env.step_policy(steps)
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(40 + case); o.reset()
        assert_records_equal(got[:, r], o.run_first_fit(steps), f"{kind} case {case}: {topo} S={S} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("on_table", [True, False], ids=["table_bit_rates", "other_bit_rates"])
@pytest.mark.parametrize("case", range(3))
def test_random_request_traces_vs_oracle(case, on_table):
    """Trace replay (ongym_set_requests): per-replica request traces written on the host - exponential arrivals and holding
    times rounded to float32, uniform node pairs, bit rates from the configured table (the lean kernel replays those) or
    arbitrary ones (generic kernel, slot counts by ceil) - stepped through the fused first fit against the oracle on the same
    traces, to the end of the trace (the steps after it are flagged no-ops)."""
    from optical_networking_gym._native import REQUEST_DTYPE
    rng = np.random.default_rng(15000 + case + (50 if on_table else 0))
    topo = ["nsfnet", "cost239", "nobel-eu"][int(rng.integers(0, 3))]
    tb = golden_tables(topo)
    S = int(rng.integers(64, 280))
    table = (10, 40, 100, 400)
    B, n = 4, 500
    load = float(rng.uniform(100, 220) * S / 100)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=load, bit_rate_selection="discrete",
              bit_rates=table, auto_reset=True, episode_length=int(rng.integers(150, 300)),
              launch_power_dbm=float(rng.uniform(-2, 2)), margin=0.0)
    reqs = np.zeros((B, n), REQUEST_DTYPE)
    for r in range(B):
        at = np.cumsum(rng.exponential(10800.0 / load, n)).astype(np.float32)
        reqs[r]["arrival_time"] = at
        reqs[r]["holding_time"] = rng.exponential(10800.0, n).astype(np.float32)
        src = rng.integers(0, tb.n_nodes, n)
        dst = (src + rng.integers(1, tb.n_nodes, n)) % tb.n_nodes
        reqs[r]["source"], reqs[r]["destination"] = src, dst
        reqs[r]["bit_rate"] = rng.choice(np.array(table), n) if on_table else rng.integers(5, 500, n)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.set_requests(reqs); env.reset()
    assert env.occupancy()["lean_kernel"] == on_table
    got = env.step_policy(n + 5)
    for r in range(B):
        # every reset() draws a request of its own (the one drawn by the terminal step is dropped, qrmsa.pyx:427-504): the
        # trace ends a few steps before n; from there on the device's steps are flagged no-ops
        noop = (got[:, r]["flags"] & nat.F_NO_REQUEST) != 0
        valid = int(np.argmax(noop))
        assert noop[valid:].all() and n - 6 <= valid <= n
        o = OracleEnv(holder, replica=r)
        o.set_trace(reqs[r]); o.reset()
        want = o.run_first_fit(valid)
        assert_records_equal(got[:valid, r], want, f"trace case {case} {topo} S={S} on_table={on_table} replica {r}")
        assert int(want["terminated"].sum()) == n - valid - 0 or int(want["terminated"].sum()) == n - valid - 1


def test_sharded_batch_equals_unsharded_bit_exact():
    """A batch split over two environments with replica bases 0 and B/2 (what two ranks of bench.py / a sharded sweep
    own, `shard_bounds`) reproduces the single environment of B replicas bit for bit: per-replica statistics, grids and
    step records.  SURVEY §4 item (4): sharded == unsharded; graph_load.py:361-363 fans out the same way."""
    from optical_networking_gym._dist import shard_bounds
    tb = golden_tables("nsfnet")
    B, steps = 512, 1100
    kw = dict(tables=tb, modulations=jocn_modulations(), num_spectrum_resources=320, capacity=448, load=300,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000)
    whole = BatchedQRMSAEnv(batch_size=B, **kw); whole.seed(5); whole.reset()
    rec_whole = whole.step_policy(steps)
    st_whole = whole.stats()
    parts, recs = [], []
    for rank in range(2):
        base, n = shard_bounds(B, rank, 2)
        e = BatchedQRMSAEnv(batch_size=n, **kw); e.seed(5, replica_base=base); e.reset()
        recs.append(e.step_policy(steps))
        parts.append(e)
    st_parts = np.concatenate([e.stats() for e in parts])
    assert st_parts.tobytes() == st_whole.tobytes()
    assert np.concatenate(recs, axis=1).tobytes() == rec_whole.tobytes()
    for r in (0, B // 2 - 1, B // 2, B - 1):
        e, lr = (parts[0], r) if r < B // 2 else (parts[1], r - B // 2)
        np.testing.assert_array_equal(e.grid(lr), whole.grid(r))
    # and the reduction the ranks do (sum of the per-rank sums) equals the unsharded sum
    for f in ("total_steps", "total_accepted", "total_gn_evals", "total_interferer_terms", "total_active_sum"):
        assert sum(int(e.stats()[f].sum()) for e in parts) == int(st_whole[f].sum())


@pytest.mark.parametrize("pid", LEAN_POLICIES)
def test_narrow_and_wide_lean_builds_agree_bit_exact(pid, monkeypatch):
    """The lean kernels exist in two builds per policy (csrc/ongym_fast.hip): the narrow one assumes every slot count of the
    configuration is <= 32 (one-step run-AND shifts, two-word marks, no wide-release path) and is what NSFNET-320 runs; the
    wide one (ONGYM_FORCE_WIDE=1 selects it for any configuration) is the general code.  Same arithmetic in the same
    order: step records, statistics and grids must be identical bit for bit.  (The wide build is held to the oracle with
    slot counts up to 80 by the *_wide_services_* tests, the narrow one by every NSFNET / COST239 / nobel-eu test.)"""
    tb = golden_tables("nsfnet")
    B, steps = 256, 1100 if pid in (nat.POLICY_FIRST_FIT, nat.POLICY_LOAD_BALANCING) else 400
    kw = dict(tables=tb, modulations=jocn_modulations(), num_spectrum_resources=320, capacity=448, load=320,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000, batch_size=B)
    outs = []
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("ONGYM_FORCE_WIDE", "1")       # read at ongym_create
        e = BatchedQRMSAEnv(**kw); e.seed(11); e.reset()
        assert e.occupancy(pid)["lean_kernel"]
        rec = e.step_policy(steps, policy=pid)
        outs.append((rec.tobytes(), e.stats().tobytes(), [e.grid(r).tobytes() for r in (0, 17, B - 1)]))
    assert outs[0][0] == outs[1][0], "step records differ between the narrow and the wide build"
    assert outs[0][1] == outs[1][1], "statistics differ between the narrow and the wide build"
    assert outs[0][2] == outs[1][2], "grids differ between the narrow and the wide build"


@pytest.mark.parametrize("generic", [False, True], ids=["lean", "generic"])
def test_lowest_fragmentation_score_tie_keeps_the_lower_route(generic, monkeypatch):
    """Found by the round-3 soak (COST239, S = 160, lowest fragmentation, replica 54 of the case, request 618): routes 2 and 4 of
    the request have mean link entropies one ulp apart and equal cuts / rss, so 0.33 * se + 0.33 * cuts + 0.34 * rss TIES when every
    product and sum is rounded on its own (Python floats: heuristics.py:375-384, the lower route index keeps the lead, :404-406) and
    differs by an ulp when `0.33 * se + 0.33 * cuts` is one fma - which is what HIP's default -ffp-contract=fast made of it in
    both kernels (__dmul_rn / __dadd_rn are plain operators).  The oracle (volatile temporaries) was right; the kernels now
    round like it (fp_barrier).  The same stream and parameters, both kernels, against the oracle to the end of the episode."""
    if generic:
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
    B, r, steps = 55, 54, 700
    loads = np.full(B, 74.35965007739468); lps = np.full(B, -0.41180015260154335); margins = np.zeros(B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=160, capacity=1024, episode_length=1000, auto_reset=True,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000), replica_load=loads,
              replica_launch_power_dbm=lps, replica_margin=margins)
    tb = golden_tables("cost239")
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw); env.seed(2025); env.reset()
    assert env.occupancy(nat.POLICY_LOWEST_FRAGMENTATION)["lean_kernel"] == (not generic)
    got = env.step_policy(steps, policy=nat.POLICY_LOWEST_FRAGMENTATION)[:, r]
    o = OracleEnv(nat.ConfigHolder(tb, batch=B, **kw), replica=r); o.seed(2025); o.reset()
    want = o.run_policy(nat.POLICY_LOWEST_FRAGMENTATION, steps)
    assert (int(want["route"][618]), int(want["modulation"][618]), int(want["slot"][618])) == (2, 4, 69)    # the tie: route 2, not 4
    assert_records_equal(got, want, "lowest-fragmentation score tie")
    np.testing.assert_array_equal(env.grid(r), o.grid())


def test_lean_launch_longer_than_the_32_bit_sum_allows_is_split_on_the_host():
    """The lean kernels add up the running services of their steps in 32 bits; `fast_launch` (csrc/ongym_fast.hip) therefore
    splits a launch so that steps x capacity stays below 2^32.  With capacity 8192 the limit is 524 287 steps: one call of
    524 287 + 60 steps (two launches, the step records of the second one written behind the first one's) must equal the same
    steps made in two calls, bit for bit."""
    tb = golden_tables("nsfnet")
    cap = 8192
    chunk = 0xFFFFFFFF // cap
    kw = dict(tables=tb, modulations=jocn_modulations(), num_spectrum_resources=320, capacity=cap, load=300,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000, batch_size=2)
    a = BatchedQRMSAEnv(**kw); a.seed(3); a.reset()
    assert a.occupancy()["lean_kernel"]
    rec_a = a.step_policy(chunk + 60)
    b = BatchedQRMSAEnv(**kw); b.seed(3); b.reset()
    b.step_policy(chunk, record=False)
    rec_b = b.step_policy(60)
    assert rec_a[chunk:].tobytes() == rec_b.tobytes()
    assert a.stats().tobytes() == b.stats().tobytes()
    st = a.stats()
    assert int(st["total_steps"][0]) == chunk + 60 and int(st["services_processed"][0]) >= chunk + 60
    assert 100 * (chunk + 60) < int(st["total_active_sum"][0]) < 400 * (chunk + 60)       # ~ 210-290 running services per step


def test_device_generator_continuous_bit_rates_vs_oracle():
    """draw_next's randint branch (bit_rate_selection="continuous", qrmsa.pyx:1088-1089) on the DEVICE generator against
    the oracle on the same (seed, replica) streams — slot counts then come from the ceil of get_number_slots
    (qrmsa.pyx:1198-1205) instead of the discrete table."""
    B, steps = 32, 1200
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=1024, episode_length=1000,
              auto_reset=True, load=700, bit_rate_selection="continuous", bit_rate_lower_bound=25,
              bit_rate_higher_bound=400, margin=0.5)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), **kw)
    want, oracles = run_oracle_batch(holder, 99, steps, B)
    ekw = dict(kw); ekw.pop("batch")
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), batch_size=B, **ekw)
    env.seed(99)
    env.reset()
    got = env.step_policy(steps)
    assert_records_equal(got, want, "continuous")
    assert len(np.unique(got["nslots"][got["accepted"] == 1])) > 10      # really a continuum of slot counts
    st = env.stats()
    for r in (0, B - 1):
        o = oracles[r].stats()
        for f in ("services_accepted", "bit_rate_requested", "bit_rate_provisioned", "episode_bit_rate_provisioned",
                  "rejected", "current_time", "active"):
            assert st[r][f] == o[f], (r, f)
        assert env.request(r).tobytes() == oracles[r].request().tobytes()
        np.testing.assert_array_equal(env.grid(r), oracles[r].grid())


def test_counters_only_reset_vs_reference_and_oracle():
    """ongym_reset_episode_counters = reset(options={"only_episode_counters": True}) (qrmsa.pyx:427-464): replay of the
    reference's captured run, then the same call in the middle of device-generated traffic vs the oracle, including replicas
    that are NOT reset (mask).  Needs cfg.track_service_ids (the id-tracking kernels)."""
    from test_oracle_golden import _epreset_checks
    meta, d = load_traj("traj_nsfnet320_epreset")
    with pytest.raises(OngymError):          # ids are needed (core/osnr.pyx:65 skips interferers by service id)
        make_env(meta, auto_reset=False).reset_episode_counters()
    env = make_env(meta, auto_reset=False, track_service_ids=True)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    recs = env.step_policy(meta["reset_at"])[:, 0]
    env.reset_episode_counters()
    st0 = env.stats()[0]
    assert st0["episode_services_processed"] == 0 and st0["episode_services_accepted"] == 0 and st0["active"] == recs["active"][-1]
    recs = np.concatenate([recs, env.step_policy(meta["term_at"] - meta["reset_at"])[:, 0]])
    st = env.stats()[0]
    env.reset()
    recs = np.concatenate([recs, env.step_policy(meta["after_full"])[:, 0]])
    _epreset_checks(meta, d, recs, st, meta["reset_at"])
    # device generator (lean kernel where eligible) vs oracle, counters reset on every second replica only
    B, kw = 16, dict(modulations=jocn_modulations(), num_spectrum_resources=320, capacity=1024, episode_length=500,
                     auto_reset=True, load=320, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400))
    holder = nat.ConfigHolder(golden_tables("nsfnet"), batch=B, **kw)
    dev = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), batch_size=B, track_service_ids=True, **kw)
    dev.seed(31); dev.reset()
    mask = (np.arange(B) % 2 == 0).astype(np.uint8)
    got = [dev.step_policy(180)]
    dev.reset_episode_counters(mask)
    got.append(dev.step_policy(700))
    got = np.concatenate(got)
    sd = dev.stats()
    for r in range(B):
        o = OracleEnv(holder, replica=r); o.seed(31); o.reset()
        want = [o.run_first_fit(180)]
        if mask[r]:
            o.reset_counters()
        want.append(o.run_first_fit(700))
        assert_records_equal(got[:, r], np.concatenate(want), f"replica {r}")
        so = o.stats()
        for f in ("episode_services_processed", "episode_services_accepted", "rejected", "episode_bit_rate_requested",
                  "episode_bit_rate_provisioned", "bit_rate_requested", "active", "episodes_completed",
                  "last_episode_accepted", "last_episode_service_blocking_rate"):
            assert sd[r][f] == so[f], (r, f)
        assert sd[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        np.testing.assert_array_equal(dev.grid(r), o.grid())


def test_nonuniform_attenuation_vs_oracle():
    """per-link alpha (template path UNIFORM_ALPHA=false)."""
    import copy
    tb = copy.deepcopy(golden_tables("nsfnet"))
    tb.link_alpha = tb.link_alpha * np.linspace(0.9, 1.2, tb.n_links)
    B, steps = 8, 600
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=1024, load=300,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True)
    holder = nat.ConfigHolder(tb, **kw)
    want, _ = run_oracle_batch(holder, 5, steps, B)
    env = BatchedQRMSAEnv(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=320,
                          capacity=1024, load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400))
    env.seed(5); env.reset()
    assert_records_equal(env.step_policy(steps), want, "nonuniform")


def test_highest_snr_per_link_attenuation_vs_oracle():
    """heuristic_highest_snr (policy id 2) with per-link attenuation: the interferer field without the pair table."""
    import copy
    tb = copy.deepcopy(golden_tables("nsfnet"))
    tb.link_alpha = tb.link_alpha * np.linspace(0.9, 1.2, tb.n_links)
    B, steps = 4, 260
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=128, batch=B, capacity=512, load=200,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True, episode_length=200)
    holder = nat.ConfigHolder(tb, **kw)
    env = BatchedQRMSAEnv(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=128,
                          capacity=512, load=200, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
                          episode_length=200, auto_reset=True)
    env.seed(12); env.reset()
    got = env.step_policy(steps, policy=nat.POLICY_HIGHEST_SNR)
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(12); o.reset()
        assert_records_equal(got[:, r], o.run_policy(nat.POLICY_HIGHEST_SNR, steps), f"highest SNR, per-link alpha, replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())


def test_policy_actions_matches_step_and_is_pure():
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta, batch=16, load=450)
    env.seed(3); env.reset()
    env.step_policy(300, record=False)
    before = env.grid(5).copy()
    acts, flags = env.policy_actions()
    np.testing.assert_array_equal(env.grid(5), before)
    rec = env.step(acts)
    assert np.array_equal(rec["action"], acts)
    assert not rec["retry"].any() and not (rec["flags"] & nat.F_QOT_ERROR).any()
    assert np.array_equal(rec["accepted"] == 0, acts == env.reject_action)


def test_qot_infeasible_action_is_flagged_not_applied():
    """The reference raises ValueError (qrmsa.pyx:925-929): 64QAM over a 3450 km path fails QoT on an empty network."""
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta)
    reqs = traj_requests(d)[:4].copy()
    reqs["source"], reqs["destination"], reqs["bit_rate"] = 0, 12, 400.0
    env.set_requests(reqs)
    env.reset()
    rec = env.step(np.array([0], np.int32))[0]     # path 0, relative modulation 0 = 64QAM, slot 0
    assert rec["flags"] & nat.F_QOT_ERROR and not rec["accepted"]
    assert env.services(0).size == 0 and env.grid(0).all()
    assert env.request(0)["bit_rate"] == 400.0 and env.stats()[0]["total_steps"] == 0


def test_masked_reset_and_capacity_overflow():
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta, batch=4, capacity=64, load=600)
    env.seed(9); env.reset()
    env.step_policy(400, record=False)
    with pytest.raises(OngymError):
        env.stats()                                   # 64 concurrent services cannot hold load 600
    env = make_env(meta, batch=4, load=300)
    env.seed(9); env.reset()
    env.step_policy(200, record=False)
    g1 = env.grid(1).copy()
    env.reset(np.array([1, 0, 1, 0], np.uint8))
    assert env.grid(0).all() and env.grid(2).all()
    np.testing.assert_array_equal(env.grid(1), g1)
    st = env.stats()
    assert st[0]["episode_services_processed"] == 1 and st[1]["episode_services_processed"] == 201


def check_state_invariants(env, tables, replica, S):
    """grid == complement of the union of the running services' [slot, slot+n(+1 guard)) on their links."""
    svc = env.services(replica)
    grid = np.ones((tables.n_links, S), np.int32)
    for s in svc:
        end = s["slot"] + s["nslots"]
        end = end + 1 if end < S else end
        for l in tables.path_links[s["path_id"]][:tables.path_hops[s["path_id"]]]:
            assert grid[l, s["slot"]:end].all(), "overlapping allocations"
            grid[l, s["slot"]:end] = 0
    np.testing.assert_array_equal(env.grid(replica), grid)


def test_full_batch_invariants_nsfnet_4096():
    """BASELINE config 2 size (B=4096): size-independent properties — determinism, state consistency, conservation."""
    tb = golden_tables("nsfnet")
    kw = dict(tables=tb, modulations=jocn_modulations(), batch_size=4096, num_spectrum_resources=320, capacity=512,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000)
    a = BatchedQRMSAEnv(**kw); a.seed(11); a.reset(); a.step_policy(1500, record=False)
    b = BatchedQRMSAEnv(**kw); b.seed(11); b.reset()
    for _ in range(3):
        b.step_policy(500, record=False)             # launch partition must not matter
    sa, sb = a.stats(), b.stats()
    assert sa.tobytes() == sb.tobytes()
    assert (sa["total_steps"] == 1500).all() and (sa["episodes_completed"] == 1).all()
    assert (sa["services_processed"] == 1502).all()  # 1500 steps + initial request + the one dropped at the boundary
    assert (sa["episode_services_accepted"] + sa["rejected"] == sa["episode_services_processed"] - 1).all()
    blocking = 1 - sa["total_accepted"].sum() / sa["total_steps"].sum()
    assert 0.0 <= blocking < 0.05                     # reference: 0.0187 at load 300 in steady state
    for r in (0, 1234, 4095):
        check_state_invariants(a, tb, r, 320)
        assert len(a.services(r)) == sa[r]["active"]
    c = BatchedQRMSAEnv(**kw); c.seed(12); c.reset(); c.step_policy(200, record=False)
    assert c.stats()["services_accepted"].tobytes() != sa["services_accepted"].tobytes()


def test_bench_size_batch_65536_properties_and_sampled_oracle():
    """The bench configuration itself (NSFNET-320, B = 65 536, capacity 448, 250 steps per launch): no overflow, launch
    partition independence and conservation over the whole batch; state consistency and an oracle replay of the same
    (seed, replica) streams on sampled replicas."""
    tb = golden_tables("nsfnet")
    B, steps = 65536, 1250
    kw = dict(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=320, capacity=448,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000)
    a = BatchedQRMSAEnv(**kw); a.seed(1); a.reset()
    for _ in range(steps // 250):
        a.step_policy(250, record=False)
    sa = a.stats()
    assert not (sa["flags"] & nat.F_OVERFLOW).any() and sa["active"].max() < 448
    assert (sa["total_steps"] == steps).all() and (sa["episodes_completed"] == 1).all()
    assert (sa["episode_services_accepted"] + sa["rejected"] == sa["episode_services_processed"] - 1).all()
    assert 0.005 < 1 - sa["total_accepted"].sum() / sa["total_steps"].sum() < 0.02
    b = BatchedQRMSAEnv(**kw); b.seed(1); b.reset(); b.step_policy(steps, record=False)
    assert b.stats().tobytes() == sa.tobytes()
    holder = nat.ConfigHolder(tb, modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=448,
                              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000,
                              auto_reset=True)
    rng = np.random.default_rng(7)
    for r in [0, 1, 31337, B - 1] + [int(x) for x in rng.integers(2, B - 1, 44)]:   # 48 replicas: first/last blocks + random
        check_state_invariants(a, tb, r, 320)
        o = OracleEnv(holder, replica=r)
        o.seed(1); o.reset(); o.run_first_fit(steps)
        so = o.stats()
        for f in ("services_accepted", "episode_services_accepted", "rejected", "bit_rate_provisioned", "active",
                  "current_time", "last_episode_accepted", "last_service_blocking_rate", "total_paths_tried"):
            assert sa[r][f] == so[f], (r, f)
        assert sa[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        np.testing.assert_array_equal(a.grid(r), o.grid())


@pytest.mark.parametrize("topo,S,load,capacity,B,blocks_per_cu,nsample",
                         [("cost239", 320, 400, 512, 16384, 18, 40),       # BASELINE config 3 as bench.py --workload cost239_320 times it
                          ("nobel-eu", 768, 600, 704, 65536, 11, 32)])     # BASELINE config 4 (bench.py --workload nobeleu768)
def test_bench_shape_c3_c4_sampled_oracle(topo, S, load, capacity, B, blocks_per_cu, nsample):
    """The kernels the C3 / C4 bench lines time, at the bench's own shape (capacity, batch, record=False, 250-step
    launches): the record-free lean instantiation (for nobel-eu the M64 codec) is held to the oracle on sampled replicas —
    counters, clocks, mean GSNR of the finished episode and the whole grid.  Semantics: envs/qrmsa.pyx:838-1122,
    heuristics/heuristics.py:923-966."""
    tb = golden_tables(topo)
    steps = 1250
    kw = dict(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=S, capacity=capacity,
              load=load, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000)
    a = BatchedQRMSAEnv(**kw); a.seed(1); a.reset()
    occ = a.occupancy()
    assert occ["lean_kernel"]
    assert occ["blocks_per_cu"] == blocks_per_cu               # what profiles/r03_*_bench.json report for this workload
    for _ in range(steps // 250):
        a.step_policy(250, record=False)
    sa = a.stats()
    assert not (sa["flags"] & nat.F_OVERFLOW).any() and sa["active"].max() < capacity
    assert (sa["total_steps"] == steps).all() and (sa["episodes_completed"] == 1).all()
    assert (sa["episode_services_accepted"] + sa["rejected"] == sa["episode_services_processed"] - 1).all()
    holder = nat.ConfigHolder(tb, modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=capacity,
                              load=load, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000,
                              auto_reset=True)
    rng = np.random.default_rng(11)
    for r in [0, 1, B - 1] + [int(x) for x in rng.integers(2, B - 1, nsample - 3)]:
        check_state_invariants(a, tb, r, S)
        o = OracleEnv(holder, replica=r)
        o.seed(1); o.reset(); o.run_first_fit(steps)
        so = o.stats()
        for f in ("services_accepted", "episode_services_accepted", "rejected", "bit_rate_provisioned", "active",
                  "current_time", "last_episode_accepted", "last_service_blocking_rate", "total_paths_tried"):
            assert sa[r][f] == so[f], (r, f)
        assert sa[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        np.testing.assert_array_equal(a.grid(r), o.grid())


@pytest.mark.parametrize("pid,B,steps,nsample", [(1, 65536, 1250, 32), (2, 16384, 750, 16), (10, 16384, 750, 16)])
def test_bench_shape_lean_policies_sampled_oracle(pid, B, steps, nsample):
    """The lean kernels of the other three JOCN heuristics (graph_load.py:116-125) at the shape `bench.py --policy` times them
    (NSFNET-320, load 300, capacity 448, record=False, 250-step launches, whole episodes from the empty network): counters,
    clocks, mean GSNR and grids of sampled replicas against the oracle's restatement of the heuristic (OpenMP over the
    sample).  Semantics: heuristics.py:547-627 (load balancing), :272-328 (highest SNR), :330-414 (lowest fragmentation)."""
    from oracle_lib import batch_run_policy
    tb = golden_tables("nsfnet")
    kw = dict(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=320, capacity=448,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000)
    a = BatchedQRMSAEnv(**kw); a.seed(1); a.reset()
    assert a.occupancy(pid)["lean_kernel"]
    for _ in range(steps // 250):
        a.step_policy(250, record=False, policy=pid)
    sa = a.stats()
    assert not (sa["flags"] & nat.F_OVERFLOW).any() and sa["active"].max() < 448
    assert (sa["total_steps"] == steps).all()
    assert (sa["episode_services_accepted"] + sa["rejected"] == sa["episode_services_processed"] - 1).all()
    holder = nat.ConfigHolder(tb, modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=448,
                              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000,
                              auto_reset=True)
    rng = np.random.default_rng(5 + pid)
    sample = [0, 1, B - 1] + [int(x) for x in rng.integers(2, B - 1, nsample - 3)]
    oracles = []
    for r in sample:
        o = OracleEnv(holder, replica=r)
        o.seed(1); o.reset()
        oracles.append(o)
    import os
    assert batch_run_policy(oracles, pid, steps, min(len(os.sched_getaffinity(0)), nsample)) == steps * nsample
    for r, o in zip(sample, oracles):
        check_state_invariants(a, tb, r, 320)
        so = o.stats()
        for f in ("services_accepted", "episode_services_accepted", "rejected", "bit_rate_provisioned", "active",
                  "current_time", "last_episode_accepted", "last_service_blocking_rate"):
            assert sa[r][f] == so[f], (pid, r, f)
        np.testing.assert_array_equal(sa[r]["episode_modulation_hist"], so["episode_modulation_hist"])
        if steps >= 1000:
            assert sa[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        np.testing.assert_array_equal(a.grid(r), o.grid())


@pytest.mark.parametrize("generic", [False, True], ids=["lean", "generic"])
@pytest.mark.parametrize("tag", ["traj_nsfnet320_lb", "traj_nobeleu320_lb", "traj_nsfnet128_hsnr"])
def test_other_fused_policies_vs_reference(tag, generic, monkeypatch):
    """fused load_balancing_best_modulation (heuristics.py:547-627) / heuristic_highest_snr (:272-328) against the
    reference's captured runs, through the lean kernels (k_fast<..., POL>, csrc/ongym_fast.hpp) and the generic k_run."""
    meta, d = load_traj(tag)
    pid = {"load_balancing": nat.POLICY_LOAD_BALANCING, "highest_snr": nat.POLICY_HIGHEST_SNR}[meta["policy"]]
    if generic:
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
    env = make_env(meta, auto_reset=True)
    env.set_requests(traj_requests(d))
    assert env.occupancy(pid)["lean_kernel"] == (not generic)
    for _ in range(meta["initial_resets"]):
        env.reset()
    rec = env.step_policy(meta["n_steps"], policy=pid)[:, 0]
    for f, g in (("action", "st_action"), ("accepted", "st_accepted"), ("terminated", "st_term"), ("active", "st_active"),
                 ("route", "st_route"), ("slot", "st_slot"), ("reward", "st_reward")):
        assert np.array_equal(rec[f], d[g]), f
    assert np.array_equal((rec["flags"] & nat.F_BLOCKED_RESOURCES) != 0, d["st_bres"] == 1)
    assert np.array_equal((rec["flags"] & nat.F_BLOCKED_OSNR) != 0, d["st_bosnr"] == 1)
    np.testing.assert_allclose(rec["osnr"], d["st_osnr"], rtol=GSNR_RTOL)
    assert len(np.unique(rec["route"][rec["accepted"] == 1])) > 1      # both policies do spread over routes


@pytest.mark.parametrize("pid,S,load,steps", [(1, 320, 550, 900), (2, 96, 90, 500)])
def test_other_policies_vs_oracle_random_traffic(pid, S, load, steps):
    B = 32
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=1024, load=load,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True)
    holder = nat.ConfigHolder(golden_tables("cost239"), **kw)
    want = np.zeros((steps, B), nat.STEP_DTYPE)
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(99); o.reset()
        want[:, r] = o.run_policy(pid, steps)
    env = BatchedQRMSAEnv(tables=golden_tables("cost239"), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=S, capacity=1024, load=load, bit_rate_selection="discrete",
                          bit_rates=(10, 40, 100, 400))
    env.seed(99); env.reset()
    assert_records_equal(env.step_policy(steps, policy=pid), want, f"policy {pid}")
    acts, flags = env.policy_actions(policy=pid)
    rec = env.step(acts)
    assert not rec["retry"].any() and not (rec["flags"] & nat.F_QOT_ERROR).any()


def test_extreme_shapes_vs_oracle(tmp_path):
    """limits of the kernels: 40-node ring (only 2 of k=5 paths exist per pair, up to 39 hops = 39 lanes per path),
    S = 1000 slots (16 bitmap words, not a multiple of 64), 1 Tb/s requests (80 slots > one word), 40 links (> 32: generic
    record codec)."""
    from optical_networking_gym._tables import StaticTables
    from optical_networking_gym.topology import get_topology
    n = 40
    lines = [str(n), str(n)] + [f"{i + 1} {(i + 1) % n + 1} {150 + 10 * (i % 7)}" for i in range(n)]
    f = tmp_path / "ring40.txt"
    f.write_text("\n".join(lines) + "\n")
    topo = get_topology(str(f), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    tables = StaticTables.from_topology(topo)
    assert tables.max_hops == 39 and (tables.pair_paths[:, :, 2:] == -1).all() and tables.n_links == 40
    B, steps = 8, 500
    kw = dict(num_spectrum_resources=1000, capacity=1024, load=250, bit_rate_selection="discrete",
              bit_rates=(10, 100, 400, 1000), episode_length=400)
    holder = nat.ConfigHolder(tables, modulations=jocn_modulations(), batch=B, auto_reset=True, **kw)
    want, oracles = run_oracle_batch(holder, 4, steps, B)
    env = BatchedQRMSAEnv(tables=tables, modulations=jocn_modulations(), batch_size=B, **kw)
    env.seed(4); env.reset()
    got = env.step_policy(steps)
    assert_records_equal(got, want, "ring40/S1000")
    assert got["nslots"].max() >= 40 and (got["accepted"] == 0).any() and got["slot"].max() > 900
    np.testing.assert_array_equal(env.grid(3), oracles[3].grid())


def test_measure_disruptions_vs_reference_and_oracle():
    """measure_disruptions=True (qrmsa.pyx:937-952): per-episode disrupted counts against the reference's captured run,
    then counts and the membership of the disrupted list against the oracle on device-generated traffic."""
    meta, d = load_traj("traj_nsfnet320_disr")
    env = make_env(meta, auto_reset=True, measure_disruptions=True)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    o = OracleEnv(holder_for(meta, measure_disruptions=True))
    o.set_trace(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        o.reset()
    done = 0
    for chunk in (400, 598, 1, 600, 399):                       # 999 steps = one episode
        rec = env.step_policy(chunk)[:, 0]
        assert np.array_equal(rec["action"], d["st_action"][done:done + chunk])
        for _ in range(chunk):
            a, _, _ = o.policy_first_fit()
            rc, r = o.step(a)
            if r["terminated"]:
                o.reset()
        done += chunk
        s, so = env.stats()[0], o.stats()
        for f in ("disrupted_services", "episode_disrupted_services", "last_episode_disrupted", "services_accepted"):
            assert s[f] == so[f], (done, f, s[f], so[f])
        # the reference's info["disrupted_services"] at the last step of the chunk
        assert abs(s["disrupted_services"] - d["st_disr"][done - 1] * s["services_accepted"]) < 1e-6 or rec["terminated"][-1]
    assert env.stats()[0]["last_episode_disrupted"] > 5
    # random traffic, per-replica power so that some replicas see many disruptions and some none
    B, steps = 24, 700
    lps = np.linspace(-2.0, 5.0, B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=1024, load=500,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True,
              replica_launch_power_dbm=lps, measure_disruptions=True)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=320, capacity=1024, load=500, bit_rate_selection="discrete",
                          bit_rates=(10, 40, 100, 400), replica_launch_power_dbm=lps, measure_disruptions=True)
    env.seed(12); env.reset()
    got = env.step_policy(steps)
    st = env.stats()
    for r in range(B):
        oo = OracleEnv(holder, replica=r)
        oo.seed(12); oo.reset()
        want = oo.run_first_fit(steps)
        assert np.array_equal(got["action"][:, r], want["action"])
        so = oo.stats()
        assert st[r]["disrupted_services"] == so["disrupted_services"], (r, st[r]["disrupted_services"], so["disrupted_services"])
        flagged = env.services(r)
        assert int(flagged["reserved"].sum()) <= so["disrupted_services"]      # some disrupted services have departed
    assert st["disrupted_services"].max() > 10 and len(np.unique(st["disrupted_services"])) > 5


# ---- defragmentation (qrmsa.pyx:1117-1119, 1545-1639) -----------------------------------------------------------------
@pytest.mark.parametrize("tag", ["traj_nsfnet320_defrag", "traj_nsfnet320_defrag4"])
def test_defragmentation_vs_reference(tag):
    """The reference run with defragmentation=True: decisions, grid snapshots (they move with every reallocation), the
    per-step info counters and the episode's mean GSNR (which sees the OSNR defragment() rewrites)."""
    meta, d = load_traj(tag)
    env = make_env(meta, auto_reset=True, defragmentation=True, n_defrag_services=meta["n_defrag_services"])
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    snaps = {int(s): i for i, s in enumerate(d["snap_step"])}
    cuts = sorted(set([s + 1 for s in snaps] + [57, meta["episode_length"] - 1, meta["n_steps"]]))
    done = 0
    for cut in cuts:
        rec = env.step_policy(cut - done)[:, 0]
        sl = slice(done, cut)
        for f, g in (("action", "st_action"), ("accepted", "st_accepted"), ("terminated", "st_term"), ("active", "st_active"),
                     ("route", "st_route"), ("slot", "st_slot"), ("reward", "st_reward")):
            assert np.array_equal(rec[f], d[g][sl]), (f, done)
        acc = d["st_accepted"][sl] == 1
        np.testing.assert_allclose(rec["osnr"][acc], d["st_osnr"][sl][acc], rtol=GSNR_RTOL)
        done = cut
        st = env.stats()[0]
        if not rec["terminated"][-1]:      # (auto-reset zeroes the episode counters right after a terminal step)
            assert st["step_defrag_cycles"] == d["st_dcyc"][done - 1], done
            assert st["step_service_reallocations"] == d["st_drea"][done - 1], done
        if done - 1 in snaps:
            grid = np.unpackbits(d["snap_grid"][snaps[done - 1]], axis=1, bitorder="little")[:, :meta["S"]]
            np.testing.assert_array_equal(env.grid(0), grid.astype(np.int32))
        if done == meta["episode_length"] - 1:
            ti = meta["terminal_infos"][0]
            assert st["last_mean_gsnr"] == pytest.approx(ti["mean_gsnr"], rel=1e-9)
            assert st["last_episode_defrag_cycles"] == ti["episode_defrag_cicles"]
            assert st["last_episode_service_reallocations"] == ti["episode_service_realocations"]
    st = env.stats()[0]
    ti = meta["terminal_infos"][-1]
    assert st["episodes_completed"] == meta["episodes"]
    assert st["last_mean_gsnr"] == pytest.approx(ti["mean_gsnr"], rel=1e-9)
    assert st["last_episode_service_reallocations"] == ti["episode_service_realocations"] > 20


@pytest.mark.parametrize("n_defrag,topo,S,load", [(0, "nsfnet", 320, 260), (6, "cost239", 320, 420), (3, "nobel-eu", 256, 300)])
def test_defragmentation_random_traffic_vs_oracle(n_defrag, topo, S, load):
    """device traffic generator + defragmentation against the CPU oracle: records, grids, the running services with
    their ids and rewritten OSNR, the counters; per-replica launch power; crosses an episode boundary."""
    B, steps = 12, 620
    lps = np.linspace(-2.0, 3.0, B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=1024, episode_length=400,
              auto_reset=True, load=load, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400),
              replica_launch_power_dbm=lps, defragmentation=True, n_defrag_services=n_defrag)
    holder = nat.ConfigHolder(golden_tables(topo), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables(topo), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=S, capacity=1024, episode_length=400, auto_reset=True, load=load,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), replica_launch_power_dbm=lps,
                          defragmentation=True, n_defrag_services=n_defrag)
    env.seed(77); env.reset()
    got = np.concatenate([env.step_policy(211), env.step_policy(steps - 211)])
    st = env.stats()
    moved = 0
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(77); o.reset()
        want = o.run_first_fit(steps)
        assert_records_equal(got[:, r], want, f"replica {r}")
        so = o.stats()
        for f in ("episode_defrag_cycles", "episode_service_reallocations", "step_defrag_cycles",
                  "step_service_reallocations", "last_episode_defrag_cycles", "last_episode_service_reallocations",
                  "services_accepted", "episodes_completed", "active"):
            assert st[r][f] == so[f], (r, f, st[r][f], so[f])
        assert st[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        assert st[r]["episode_osnr_sum"] == pytest.approx(so["episode_osnr_sum"], rel=1e-9)
        np.testing.assert_array_equal(env.grid(r), o.grid())
        a = np.sort(env.services(r), order="service_id")
        b = np.sort(o.services(), order="service_id")
        for f in ("service_id", "path_id", "slot", "nslots", "modulation", "release_time"):
            assert np.array_equal(a[f], b[f]), (r, f)
        np.testing.assert_allclose(a["osnr"], b["osnr"], rtol=GSNR_RTOL)
        moved += int(so["last_episode_service_reallocations"])
    assert moved > 50


def test_c_abi_rejects_bad_arguments_without_touching_the_device():
    """Every operand of a query / step is validated on the host before a launch (include/ongym.h: errors are codes +
    ongym_last_error, nothing throws, nothing out of range reaches a kernel)."""
    import copy
    meta, d = load_traj("traj_nsfnet320")
    env = make_env(meta, batch=2)
    with pytest.raises(OngymError, match="no request source"):
        env.reset()
    env.seed(3); env.reset(); env.step_policy(50, record=False)
    P = golden_tables("nsfnet").n_paths
    for bad in ([(P, 0, 4)], [(-1, 0, 4)], [(0, -1, 4)], [(0, 0, 0)], [(0, 318, 4)], [(0, 0, 4), (0, 320, 1)]):
        with pytest.raises(OngymError, match="out of"):
            env.gsnr_many(0, bad)
    with pytest.raises(OngymError, match="replica out of range"):
        env.gsnr_many(2, [(0, 0, 4)])
    assert env.gsnr_many(0, np.zeros((0, 3), np.int32)).shape == (0, 3)
    with pytest.raises(OngymError, match="replica out of range"):
        env.grid(5)
    with pytest.raises(OngymError, match="path id out of range"):
        env.available_slots(0, P)
    with pytest.raises(OngymError, match="unknown policy"):
        env.step_policy(1, policy=99)
    with pytest.raises(OngymError, match="nsteps"):
        env.step_policy(0)
    moves, total = env.moves(0)                       # defragmentation is off: empty, not an error
    assert total == 0 and len(moves) == 0
    # create-time limits
    tb = copy.deepcopy(golden_tables("nsfnet"))
    tb.link_alpha = tb.link_alpha.copy(); tb.link_alpha[3] *= 1.1
    for kw in (dict(defragmentation=True), dict(measure_disruptions=True)):
        with pytest.raises(OngymError, match="uniform attenuation"):
            BatchedQRMSAEnv(tables=tb, modulations=jocn_modulations(), batch_size=1, load=100,
                            bit_rate_selection="discrete", bit_rates=(10, 40), **kw)
    with pytest.raises(OngymError, match="n_defrag_services"):
        make_env(meta, defragmentation=True, n_defrag_services=-1)


@pytest.mark.parametrize("pid", [3, 4, 5, 6, 7, 8, 9])
def test_misc_fused_policies_vs_oracle_random_traffic(pid):
    """policy ids 3..9 (lowest spectrum, LB first fit, best-modulation LB, simplified / sequential MSCL, PSR, exact fit)
    driving whole batched episodes against the oracle's restatements (which tests/test_oracle_golden.py pins to
    decisions captured from the reference).  Exact fit exercises the occupied-slots penalty inside the fused loop."""
    B, steps = 10, 520
    loads = np.linspace(350, 800, B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=256, batch=B, capacity=1024, episode_length=400,
              auto_reset=True, load=500, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), replica_load=loads)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=256, capacity=1024, episode_length=400, auto_reset=True, load=500,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), replica_load=loads)
    env.seed(5); env.reset()
    got = env.step_policy(steps, policy=pid)
    st = env.stats()
    rejected = retried = 0
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(5); o.reset()
        want = o.run_policy(pid, steps)
        assert_records_equal(got[:, r], want, f"policy {pid} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())
        assert st[r]["services_accepted"] == o.stats()["services_accepted"]
        rejected += int((want["accepted"] == 0).sum()); retried += int(want["retry"].sum())
    assert rejected > 5
    if pid == 9:
        assert retried > 50


@pytest.mark.parametrize("pid,S,B,warm,steps,loads", [(10, 96, 5, 300, 200, (150, 330)), (11, 64, 4, 400, 120, (120, 260))])
def test_scored_fused_policies_vs_oracle_random_traffic(pid, S, B, warm, steps, loads, monkeypatch):
    """policy ids 10 (lowest fragmentation) and 11 (full MSCL) driving whole batched episodes (csrc/ongym_scored.hpp)
    against the oracle's restatements, which tests/test_oracle_golden.py pins to decisions captured from the reference
    (dec_nsfnet96_lf, dec_nsfnet64_mscl), after a first-fit warm-up that fills the network.  Bit-exact records: the float
    score of lowest fragmentation decides by strict `<` and has to be reproduced to the last bit.  About a third of the
    lowest-fragmentation decisions fail the step's own GSNR check (sized slots + 1, provisioned at slots: 4x the NLI) - the
    reference raises ValueError there (50 of the fixture's 140 decisions); the fused loop rejects and flags them."""
    loads = np.linspace(loads[0], loads[1], B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=512, episode_length=1000,
              auto_reset=True, load=100, bit_rate_selection="discrete", bit_rates=(10, 40, 100), replica_load=loads)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=S, capacity=512, episode_length=1000, auto_reset=True, load=100,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100), replica_load=loads)
    env.seed(9); env.reset()
    assert env.occupancy(pid)["lean_kernel"] == (pid in LEAN_POLICIES)
    env.step_policy(warm, record=False)
    got = env.step_policy(steps, policy=pid)
    if pid in LEAN_POLICIES:         # ... and the generic kernel gives the same records
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
        env_g = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=B,
                                num_spectrum_resources=S, capacity=512, episode_length=1000, auto_reset=True, load=100,
                                bit_rate_selection="discrete", bit_rates=(10, 40, 100), replica_load=loads)
        env_g.seed(9); env_g.reset()
        assert not env_g.occupancy(pid)["lean_kernel"]
        env_g.step_policy(warm, record=False)
        assert_records_equal(env_g.step_policy(steps, policy=pid), got, "generic vs lean")
    st = env.stats()
    rejected = qot = 0
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(9); o.reset()
        o.run_policy(0, warm)
        want = o.run_policy(pid, steps)
        qot += int(((want["flags"] & nat.F_QOT_ERROR) != 0).sum())
        assert_records_equal(got[:, r], want, f"policy {pid} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())
        assert st[r]["services_accepted"] == o.stats()["services_accepted"]
        rejected += int((want["accepted"] == 0).sum())
    assert rejected > 5 and (qot > 50 if pid == 10 else qot == 0)
    # the policy alone (no step) returns the same decision the fused loop is about to apply
    acts, flags = env.policy_actions(policy=pid)
    rec = env.step(acts)
    ok = (rec["flags"] & nat.F_QOT_ERROR) == 0
    assert np.array_equal(rec["action"][ok], acts[ok])


def test_long_wide_sweep_final_state_vs_oracle():
    """256 replicas x 3 000 steps (three episodes) with launch power -7..+7 dBm, load 120..900 Erlang and margins 0..2 dB
    spread over the replicas: every replica's final grid, clocks and counters against the oracle (OpenMP over replicas).
    A single differing accept/slot decision anywhere in the 768 000 steps would show up in these."""
    from oracle_lib import batch_run_first_fit
    B, steps = 256, 3000
    rng = np.random.default_rng(42)
    loads = rng.uniform(120, 900, B)
    lps = rng.uniform(-7.0, 7.0, B)
    margins = rng.choice([0.0, 0.25, 1.0, 2.0], B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=320, batch=B, capacity=1024, episode_length=1000,
              auto_reset=True, load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000),
              replica_load=loads, replica_launch_power_dbm=lps, replica_margin=margins)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=320, capacity=1024, episode_length=1000, auto_reset=True, load=300,
                          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000), replica_load=loads,
                          replica_launch_power_dbm=lps, replica_margin=margins)
    env.seed(99); env.reset()
    for _ in range(steps // 500):
        env.step_policy(500, record=False)
    st = env.stats()
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(99); o.reset()
        oracles.append(o)
    assert batch_run_first_fit(oracles, steps, 16) == B * steps
    blocked = 0
    for r, o in enumerate(oracles):
        so = o.stats()
        for f in ("services_processed", "services_accepted", "episode_services_accepted", "rejected",
                  "bit_rate_provisioned", "episode_bit_rate_provisioned", "episodes_completed", "current_time", "active",
                  "last_episode_accepted", "last_service_blocking_rate", "last_bit_rate_blocking_rate",
                  "total_paths_tried", "total_path_hops", "total_active_sum"):
            assert st[r][f] == so[f], (r, f, st[r][f], so[f], loads[r], lps[r], margins[r])
        np.testing.assert_array_equal(st[r]["last_modulation_hist"], so["last_modulation_hist"])
        assert st[r]["last_mean_gsnr"] == pytest.approx(so["last_mean_gsnr"], rel=1e-9)
        np.testing.assert_array_equal(env.grid(r), o.grid())
        blocked += int(so["total_steps"] - so["total_accepted"])
    assert blocked > 50000          # the sweep does reach heavily blocked regimes


_STREAM_SCRIPT = r'''
import sys
sys.path[:0] = [%r, %r]
import numpy as np
from common import golden_tables, jocn_modulations
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv
import torch
tb = golden_tables("nsfnet")
B, steps = 256, 40
kw = dict(tables=tb, modulations=jocn_modulations(), batch_size=B, num_spectrum_resources=320, capacity=448, load=300,
          bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), episode_length=1000, io_device=True)
dev = torch.device("cuda", 0)
out = []
for shared in (False, True):
    env = BatchedQRMSAEnv(**kw)
    c = env.holder.struct
    obs = torch.empty((B, 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12), dtype=torch.float32, device=dev)
    mask = torch.empty((B, c.k_paths * c.n_mods_consider * c.n_slots + 1), dtype=torch.uint8, device=dev)
    actions = torch.empty(B, dtype=torch.int32, device=dev)
    recs = torch.empty((steps, B, nat.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    env.seed(3); env.reset()
    env.step_policy(300, record=False); env.sync()
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        if shared:
            env.set_stream(stream.cuda_stream)
        for i in range(steps):
            env._check(env.lib.ongym_observe(env._h, obs.data_ptr(), mask.data_ptr()), "observe")
            if not shared:
                env.sync()
            # the lowest allowed action of every replica (deterministic), computed by torch on the same stream
            actions.copy_(torch.argmax(mask, dim=1))
            if not shared:
                stream.synchronize()
            env._check(env.lib.ongym_step_actions(env._h, actions.data_ptr(), recs[i].data_ptr()), "step")
            if not shared:
                env.sync()
        stream.synchronize()
        if shared:
            env.set_stream(None)
    out.append(recs.cpu().numpy().view(nat.STEP_DTYPE).reshape(steps, B).copy())
    assert out[-1]["accepted"].mean() > 0.5
for f in nat.STEP_DTYPE.names:       # (field by field: the records' padding bytes are not written)
    assert np.array_equal(out[0][f], out[1][f]), f
# ... and on the DEFAULT stream (handle 0), which is what torch.cuda.current_stream() is unless the caller changed it
env = BatchedQRMSAEnv(**kw)
env.seed(3); env.reset(); env.step_policy(300, record=False); env.sync()
assert torch.cuda.current_stream().cuda_stream == 0
env.set_stream(torch.cuda.current_stream().cuda_stream)
c = env.holder.struct
obs = torch.empty((B, 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12), dtype=torch.float32, device=dev)
mask = torch.empty((B, c.k_paths * c.n_mods_consider * c.n_slots + 1), dtype=torch.uint8, device=dev)
actions = torch.empty(B, dtype=torch.int32, device=dev)
recs = torch.empty((steps, B, nat.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
for i in range(steps):
    env._check(env.lib.ongym_observe(env._h, obs.data_ptr(), mask.data_ptr()), "observe")
    actions.copy_(torch.argmax(mask, dim=1))
    env._check(env.lib.ongym_step_actions(env._h, actions.data_ptr(), recs[i].data_ptr()), "step")
torch.cuda.synchronize()
got = recs.cpu().numpy().view(nat.STEP_DTYPE).reshape(steps, B)
for f in nat.STEP_DTYPE.names:
    assert np.array_equal(got[f], out[0][f]), f
print("stream test ok")
'''


def test_caller_stream_equals_own_stream(tmp_path):
    """ongym_set_stream: observe -> torch ops -> step on torch's current stream with NO host synchronisation in between gives
    the records of the same loop run on the environment's own stream with a synchronisation after every call.  Runs in a fresh
    process: PyTorch brings its own copy of the HIP runtime, which refuses to initialise late in a process that has already
    created and destroyed a few hundred streams through the system's one."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "stream_test.py"
    script.write_text(_STREAM_SCRIPT % (os.path.join(repo, "optical-networking-gym_amd"), os.path.join(repo, "tests")))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "stream test ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_action_above_the_reject_index_wraps_like_the_reference():
    """encoded_decimal_to_array (qrmsa.pyx:801-834) decodes by modulo: an action index above the reject action aliases a
    (route, format, slot) - the heuristics produce such indices under a narrow codec (route K-1 with a format more than
    modulations_to_consider - 1 below max_modulation_idx, heuristics.py:36-54).  Device step vs the oracle's step in lockstep
    on nobel-eu at low launch power (long routes force low formats), both for hand-made indices and for the fused first fit
    under a narrow codec; at least one index above the reject action must occur."""
    B, Mc, S, steps = 8, 2, 160, 220
    tb = golden_tables("nobel-eu")
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=250, bit_rate_selection="discrete",
              bit_rates=(10, 40, 100, 400), auto_reset=True, modulations_to_consider=Mc, launch_power_dbm=-6.0)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(5); env.reset()
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r); o.seed(5); o.reset()
        oracles.append(o)
    reject = env.reject_action
    above = 0
    for t in range(steps):
        acts, _ = env.policy_actions()
        for r, o in enumerate(oracles):
            assert acts[r] == o.policy_first_fit()[0], (t, r)
        if t % 5 == 4:
            acts = acts.copy()
            acts[t % B] = reject + 1 + (7 * t) % (3 * S)      # a hand-made index above the reject action
        above += int((acts > reject).sum())
        rec = env.step(acts)
        for r, o in enumerate(oracles):
            rc, w = o.step(int(acts[r]))
            if rc != 0:           # the reference raises on a QoT-infeasible action: flagged, nothing applied, next try is a reject
                assert rec[r]["flags"] & nat.F_QOT_ERROR, (t, r)
                rec_r = env.step(np.where(np.arange(B) == r, reject, -1).astype(np.int32))      # -1: occupied-slots penalty elsewhere
                rc2, w2 = o.step(reject)
                assert rc2 == 0
                continue
            for f in ("accepted", "route", "modulation", "slot", "nslots", "active", "retry"):
                assert rec[r][f] == w[f], (t, r, f, int(acts[r]))
    assert above >= steps // 5


def test_m64_interferers_beyond_the_register_cache_vs_oracle():
    """nobel-eu (41 links: the M64 lean kernel, whose interferer list only holds the 256 entries of the register cache) under
    a load that puts far more than 256 interferers on a route: the ones beyond the cache are found by the second scan of the
    records (eval_one, csrc/ongym_fast.hpp).  Records and grids against the oracle, first fit and load balancing."""
    tb = golden_tables("nobel-eu")
    B, steps = 6, 1400
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=768, capacity=1536, load=1500, bit_rate_selection="discrete",
              bit_rates=(10, 40, 100), auto_reset=True, episode_length=2000)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    for pid in (nat.POLICY_FIRST_FIT, nat.POLICY_LOAD_BALANCING):
        env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
        env.seed(41); env.reset()
        assert env.occupancy(pid)["lean_kernel"]
        got = env.step_policy(steps, policy=pid)
        st = env.stats()
        assert st["active"].min() > 800                      # ~45 % of them share a link with a 5-hop route
        assert (st["total_interferer_terms"] / np.maximum(st["total_gn_evals"], 1)).max() > 200
        for r in range(B):
            o = OracleEnv(holder, replica=r)
            o.seed(41); o.reset()
            assert_records_equal(got[:, r], o.run_policy(pid, steps), f"policy {pid} replica {r}")
            np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("pid", [nat.POLICY_FIRST_FIT, nat.POLICY_LOAD_BALANCING])
def test_wide_services_on_a_41_link_topology_vs_oracle(pid):
    """1 Tb/s requests (up to 80 slots: the word-loop marking and the scalar release path of the lean kernels) on nobel-eu,
    whose routes use link 31 and link 40 - the sign bits of the two link-mask words (a release of a > 32-slot service used to
    sign-extend the mask and free slots on links the service never used: found by tools/soak_vs_oracle.py in round 3).
    Loads, launch powers and margins spread over the replicas; records and final grids against the oracle."""
    tb = golden_tables("nobel-eu")
    B, steps = 32, 1800
    rng = np.random.default_rng(3)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=320, capacity=1024, episode_length=1000, auto_reset=True,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000),
              replica_load=rng.uniform(100, 900, B), replica_launch_power_dbm=rng.uniform(-6.0, 6.0, B),
              replica_margin=rng.choice([0.0, 0.5, 1.5], B))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(2025); env.reset()
    assert env.occupancy(pid)["lean_kernel"]
    got = env.step_policy(steps, policy=pid)
    assert (got["nslots"] > 32).sum() > 50                       # wide services were provisioned (and released: three episodes)
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(2025); o.reset()
        assert_records_equal(got[:, r], o.run_policy(pid, steps), f"policy {pid} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("generic", [False, True], ids=["lean", "generic"])
def test_highest_snr_exact_ties_between_equal_routes(generic, monkeypatch):
    """heuristic_highest_snr keeps the FIRST of several candidates with equal GSNR (strict `>`, heuristics.py:316).  On an empty
    network two routes of equal links give exactly equal values in the reference; the device's sum is associated differently
    and may differ in the last bits, so a later candidate only takes the lead when it is better by more than that noise
    (found by tools/soak_vs_oracle.py in round 3: launch powers / margins / loads spread over the replicas, two episode starts).
    Final states of all replicas against the oracle (OpenMP over replicas)."""
    import os
    from oracle_lib import batch_run_policy
    if generic:
        monkeypatch.setenv("ONGYM_FORCE_GENERIC", "1")
    tb = golden_tables("nsfnet")
    B, steps = 96, 1100
    rng = np.random.default_rng(11)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=160, capacity=1024, episode_length=1000, auto_reset=True,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000),
              replica_load=rng.uniform(50, 500, B), replica_launch_power_dbm=rng.uniform(-8.0, 8.0, B),
              replica_margin=rng.choice([0.0, 0.5, 1.5, 3.0], B))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(2025); env.reset()
    assert env.occupancy(nat.POLICY_HIGHEST_SNR)["lean_kernel"] == (not generic)
    got = env.step_policy(steps, policy=nat.POLICY_HIGHEST_SNR)
    st = env.stats()
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r); o.seed(2025); o.reset(); oracles.append(o)
    assert batch_run_policy(oracles, nat.POLICY_HIGHEST_SNR, steps, min(len(os.sched_getaffinity(0)), B)) == B * steps
    assert len(np.unique(got["route"][got["accepted"] == 1])) > 1
    for r, o in enumerate(oracles):
        so = o.stats()
        for f in ("services_accepted", "episode_services_accepted", "rejected", "bit_rate_provisioned", "active", "current_time"):
            assert st[r][f] == so[f], (r, f)
        np.testing.assert_array_equal(env.grid(r), o.grid())


@pytest.mark.parametrize("pid", list(range(2, 11)))
def test_policies_wide_services_41_links_vs_oracle(pid):
    """Every fused policy but full MSCL on nobel-eu (41 links) with 1 Tb/s requests (up to 80 slots) in the mix, after a
    first-fit warm-up that has provisioned and released wide services: records and grids against the oracle."""
    tb = golden_tables("nobel-eu")
    B, warm = 4, 600
    steps = 60 if pid in (2, 10) else 220
    rng = np.random.default_rng(40 + pid)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=200, capacity=1024, episode_length=1000, auto_reset=True,
              load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400, 1000),
              replica_load=rng.uniform(100, 450, B), replica_launch_power_dbm=rng.uniform(-3.0, 4.0, B),
              replica_margin=rng.choice([0.0, 0.5], B))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(77); env.reset()
    env.step_policy(warm, record=False)
    got = env.step_policy(steps, policy=pid)
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(77); o.reset(); o.run_policy(0, warm)
        assert_records_equal(got[:, r], o.run_policy(pid, steps), f"policy {pid} replica {r}")
        np.testing.assert_array_equal(env.grid(r), o.grid())
