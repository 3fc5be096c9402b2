"""Observation + action-mask kernel through the C ABI vs the reference's captured outputs and vs the
oracle on random states. Tolerance: the observation is float32; features are held to 2e-6 relative / 2e-7 absolute,
the mask bit-exact."""
import numpy as np
import pytest

from common import golden_tables, holder_for, jocn_modulations, load_traj, traj_requests
from optical_networking_gym import _native as nat
from optical_networking_gym.envs.batched import BatchedQRMSAEnv
from oracle_lib import OracleEnv
from test_gpu_parity import make_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["obs_nsfnet320", "obs_nsfnet320_dense"])
def test_observation_and_mask_vs_reference(tag):
    meta, d = load_traj(tag)
    env = make_env(meta, auto_reset=False)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    for i in range(meta["steps"] + 1):
        obs, mask = env.observe()
        want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]]
        np.testing.assert_array_equal(mask[0], want_mask, err_msg=f"mask step {i}")
        np.testing.assert_allclose(obs[0], d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            rec = env.step(np.array([d["action"][i]], np.int32))[0]
            assert not rec["retry"] and not (rec["flags"] & nat.F_QOT_ERROR)


@pytest.mark.parametrize("tag", ["obs_nsfnet320_mtc4", "obs_nsfnet160_mtc2"])
def test_modulations_to_consider_window_vs_reference(tag):
    """modulations_to_consider < len(modulations) (qrmsa.pyx:313): ongym_observe moves max_modulation_idx like
    get_max_modulation_index (:543-581) and reports the window below it (:712-717); ongym_step_actions decodes with the
    window codec (:801-834).  Reference run driven by its own mask, every step compared."""
    meta, d = load_traj(tag)
    Mc = meta["modulations_to_consider"]
    env = make_env(meta, auto_reset=False, modulations_to_consider=Mc)
    assert env.num_actions == meta["n_actions"] == 5 * Mc * meta["S"] + 1
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    for i in range(meta["steps"] + 1):
        obs, mask = env.observe()
        assert env.stats()[0]["max_modulation_idx"] == d["max_modulation_idx"][i], i
        want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]]
        np.testing.assert_array_equal(mask[0], want_mask, err_msg=f"mask step {i}")
        np.testing.assert_allclose(obs[0], d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            rec = env.step(np.array([d["action"][i]], np.int32))[0]
            assert not rec["retry"] and not (rec["flags"] & nat.F_QOT_ERROR)
            assert rec["accepted"] == d["accepted"][i]
            if rec["accepted"]:
                assert [rec["route"], rec["modulation"], rec["slot"]] == d["decoded"][i].tolist(), i


def test_observation_with_bands_vs_reference():
    """`bands` with gen_observation=True (reference driver: graph_launch_power.py:102): one slot per service (quirk Q9), the
    observation's frequencies from channel_width; k_observe and ongym_step_actions against the reference's captured run."""
    meta, d = load_traj("obs_nsfnet320_bands")
    env = make_env(meta, auto_reset=False)
    env.set_requests(traj_requests(d))
    for _ in range(meta["initial_resets"]):
        env.reset()
    for i in range(meta["steps"] + 1):
        obs, mask = env.observe()
        want_mask = np.unpackbits(d["mask"][i], bitorder="little")[:meta["n_actions"]]
        np.testing.assert_array_equal(mask[0], want_mask, err_msg=f"mask step {i}")
        np.testing.assert_allclose(obs[0], d["obs"][i], rtol=2e-6, atol=2e-7, err_msg=f"obs step {i}")
        if i < meta["steps"]:
            rec = env.step(np.array([d["action"][i]], np.int32))[0]
            assert not rec["retry"] and not (rec["flags"] & nat.F_QOT_ERROR)
            assert rec["accepted"] == d["accepted"][i]
            if rec["accepted"]:
                assert [rec["route"], rec["modulation"], rec["slot"]] == d["decoded"][i].tolist() and rec["nslots"] == 1


def test_modulations_to_consider_vs_oracle_mask_driven():
    """The window on evolving states: 10 replicas with different launch powers, each step observe -> lowest valid action of
    the mask (every 7th step the highest) -> step, device vs oracle in lockstep; then the fused first-fit heuristic's action
    index under a narrow codec equals the oracle's (heuristics.py:36-54: relative to max_modulation_idx, window-wide radix)."""
    B, Mc, S, steps = 10, 3, 192, 160
    lps = np.linspace(-6.0, 2.0, B)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=1024, load=700, bit_rate_selection="discrete",
              bit_rates=(10, 40, 100, 400), auto_reset=True, modulations_to_consider=Mc, replica_launch_power_dbm=lps)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), batch=B, **kw)
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), batch_size=B, **kw)
    env.seed(9); env.reset()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    oracles = []
    for r in range(B):
        o = OracleEnv(holder, replica=r); o.seed(9); o.reset()
        oracles.append(o)
    seen = set()
    for t in range(steps):
        obs, mask = env.observe()
        st = env.stats()
        acts = np.zeros(B, np.int32)
        for r, o in enumerate(oracles):
            want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
            assert st[r]["max_modulation_idx"] == o.max_modulation_idx, (t, r)
            seen.add(o.max_modulation_idx)
            np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"mask step {t} replica {r}")
            np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"obs step {t} replica {r}")
            valid = np.flatnonzero(want_mask[:-1])
            acts[r] = (valid[-1] if t % 7 == 6 else valid[0]) if len(valid) else env.reject_action
        rec = env.step(acts)
        for r, o in enumerate(oracles):
            rc, w = o.step(int(acts[r]))
            assert rc == 0
            for f in ("accepted", "route", "modulation", "slot", "nslots", "active", "retry"):
                assert rec[r][f] == w[f], (t, r, f)
    assert len(seen) >= 3 and obs.shape == (B, 3 + 5 + 5 * Mc * 12) and mask.shape == (B, 5 * Mc * S + 1)
    a_dev, _ = env.policy_actions()
    for r, o in enumerate(oracles):
        assert a_dev[r] == o.policy_first_fit()[0], r
    with pytest.raises(Exception):
        env.step_policy(1, policy=nat.POLICY_LOAD_BALANCING)      # only first fit is fused for a narrow codec


# capacity decides where k_observe keeps its scratch (obs_layout): 1024 -> compact (inside the release-time slots, list
# overlaid), 128 -> too small for that: everything after the state block
@pytest.mark.parametrize("topo,S,load,capacity", [("nsfnet", 320, 500, 1024), ("nobel-eu", 320, 700, 1024),
                                                   ("cost239", 192, 300, 1024), ("nsfnet", 320, 40, 128)])
def test_observation_vs_oracle_random_states(topo, S, load, capacity):
    B = 12
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, batch=B, capacity=capacity, load=load,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True)
    holder = nat.ConfigHolder(golden_tables(topo), **kw)
    env = BatchedQRMSAEnv(tables=golden_tables(topo), modulations=jocn_modulations(), batch_size=B,
                          num_spectrum_resources=S, capacity=capacity, load=load, bit_rate_selection="discrete",
                          bit_rates=(10, 40, 100, 400))
    env.seed(77); env.reset()
    env.step_policy(450, record=False)
    obs, mask = env.observe()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(77); o.reset(); o.run_first_fit(450)
        want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
        np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"mask replica {r}")
        np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"obs replica {r}")
    assert mask[:, -1].all() and mask[:, :-1].any()


@pytest.mark.parametrize("case", range(8))
def test_observation_randomised_configurations_vs_oracle(case):
    """Observation + mask on configurations drawn at random (topology, slot count, routes per pair, bit rates, load, power,
    margin), after a first-fit warm-up, against the oracle: masks bit-exact, observations within float32 rounding."""
    rng = np.random.default_rng(500 + case)
    topo = ["nsfnet", "cost239", "ring4", "nobel-eu"][int(rng.integers(0, 4))]
    tb = golden_tables(topo)
    k = int(rng.integers(1, tb.k_paths + 1))
    if k < tb.k_paths:
        tb = tb.truncated(k)
    S = int(rng.integers(40, 420))
    rates = tuple(int(x) for x in np.sort(rng.choice(np.array([10, 25, 40, 100, 200, 400]), size=int(rng.integers(1, 5)), replace=False)))
    B, warm = 5, int(rng.integers(60, 400))
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=512, load=float(rng.uniform(60, 200) * S / 100),
              bit_rate_selection="discrete", bit_rates=rates, auto_reset=True, episode_length=1000,
              launch_power_dbm=float(rng.uniform(-3, 3)), margin=float(rng.choice([0.0, 0.5, 1.0])))
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(9 + case); env.reset()
    env.step_policy(warm, record=False)
    obs, mask = env.observe()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(9 + case); o.reset(); o.run_first_fit(warm)
        want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
        np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"case {case}: {topo} S={S} k={k} rates={rates} mask replica {r}")
        np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"case {case}: obs replica {r}")


@pytest.mark.parametrize("topo,S", [("nsfnet", 320), ("nobel-eu", 320), ("cost239", 200)])
def test_observation_wide_services_vs_oracle(topo, S):
    """Observation + mask with 1 Tb/s requests (up to 80 slots: more than one bitmap word) among the running services and as
    the current request, launch powers and margins spread over the replicas, on three topologies (nobel-eu: 41 links, the
    generic record codec): against the oracle after a first-fit warm-up."""
    B, warm = 10, 520
    rng = np.random.default_rng(17)
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=S, capacity=1024, load=300, bit_rate_selection="discrete",
              bit_rates=(10, 40, 100, 400, 1000), auto_reset=True, episode_length=1000,
              replica_load=rng.uniform(150, 700, B) * S / 320, replica_launch_power_dbm=rng.uniform(-5.0, 5.0, B),
              replica_margin=rng.choice([0.0, 0.5, 1.5], B))
    holder = nat.ConfigHolder(golden_tables(topo), batch=B, **kw)
    env = BatchedQRMSAEnv(tables=golden_tables(topo), batch_size=B, **kw)
    env.seed(23); env.reset()
    env.step_policy(warm, record=False)
    obs, mask = env.observe()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    wide = 0
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(23); o.reset(); o.run_first_fit(warm)
        wide += int((o.services()["nslots"] > 32).sum())
        want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
        np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"{topo}: mask replica {r}")
        np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"{topo}: obs replica {r}")
    assert wide > 0


def test_observation_continuous_bit_rates_vs_oracle():
    """bit_rate_selection="continuous" (randint bit rates, slot counts by ceil): the observation normalises the bit rate by
    max(bit_rates) of the otherwise unused tuple (qrmsa.pyx:679, 688); device vs oracle on loaded states."""
    B = 8
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=256, capacity=1024, load=500,
              bit_rate_selection="continuous", bit_rates=(10, 40, 100), bit_rate_lower_bound=25, bit_rate_higher_bound=300,
              auto_reset=True)
    holder = nat.ConfigHolder(golden_tables("nsfnet"), batch=B, **kw)
    assert holder.struct.max_bit_rate == 100.0
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), batch_size=B, **kw)
    env.seed(21); env.reset()
    env.step_policy(380, record=False)
    obs, mask = env.observe()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(21); o.reset(); o.run_first_fit(380)
        want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
        np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"mask replica {r}")
        np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"obs replica {r}")
    assert obs[:, 0].max() > 1.0        # a 300 Gb/s request over max(bit_rates) = 100


def test_observation_per_link_attenuation_vs_oracle():
    """Per-link attenuation (a hand-built topology: the reference's loader always gives one value): no pair table, the
    field builder evaluates every (centre, interferer, shared link) asinh difference itself, the self term is summed over
    the path's links.  Device vs oracle on loaded states, masks bit-exact."""
    import copy
    tb = copy.deepcopy(golden_tables("nsfnet"))
    tb.link_alpha = tb.link_alpha * np.linspace(0.9, 1.2, tb.n_links)
    B = 6
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=256, capacity=1024, load=450,
              bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400), auto_reset=True)
    holder = nat.ConfigHolder(tb, batch=B, **kw)
    env = BatchedQRMSAEnv(tables=tb, batch_size=B, **kw)
    env.seed(31); env.reset()
    env.step_policy(420, record=False)
    obs, mask = env.observe()
    pl = np.ctypeslib.as_array(holder.struct.path_len_norm, shape=(holder.struct.n_paths,))
    for r in range(B):
        o = OracleEnv(holder, replica=r)
        o.seed(31); o.reset(); o.run_first_fit(420)
        want_obs, want_mask = o.observe(pl, holder.struct.max_bit_rate)
        np.testing.assert_array_equal(mask[r], want_mask, err_msg=f"mask replica {r}")
        np.testing.assert_allclose(obs[r], want_obs, rtol=2e-6, atol=2e-7, err_msg=f"obs replica {r}")
    assert mask[:, :-1].any() and not mask[:, :-1].all()


def test_sample_actions_uniform_over_the_mask():
    """ongym_sample_actions = gymnasium's action_space.sample(mask): always a valid action, deterministic in (seed, draw),
    uniform over the valid ones (chi-square on a replica with few valid actions), reject when nothing else is valid."""
    env = BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), batch_size=16,
                          num_spectrum_resources=320, capacity=1024, load=2000, bit_rate_selection="discrete",
                          bit_rates=(10, 40, 100, 400))
    env.seed(3); env.reset(); env.step_policy(900, record=False)
    obs, mask = env.observe()
    rng = np.random.default_rng(0)
    mask2 = mask.copy()
    mask2[0, :-1] = 0                                   # only the reject action
    keep = rng.choice(np.flatnonzero(mask[1]), size=min(11, int(mask[1].sum())), replace=False)
    mask2[1, :] = 0; mask2[1, keep] = 1                 # 11 valid actions at arbitrary positions
    mask2[2, :] = 0; mask2[2, [0, 1, 2, 3, 9597, 9598, 9599, 9600]] = 1     # head / tail bytes of the row
    a = env.sample_actions(mask2, 5, 0)
    assert np.array_equal(a, env.sample_actions(mask2, 5, 0)) and a[0] == env.reject_action
    assert mask2[np.arange(16), a].all()
    counts1, counts2 = {}, {}
    n = 3000
    for d in range(n):
        a = env.sample_actions(mask2, 5, d)
        assert mask2[np.arange(16), a].all()
        counts1[int(a[1])] = counts1.get(int(a[1]), 0) + 1
        counts2[int(a[2])] = counts2.get(int(a[2]), 0) + 1
    for counts, k in ((counts1, len(keep)), (counts2, 8)):
        assert len(counts) == k
        chi2 = sum((c - n / k) ** 2 / (n / k) for c in counts.values())
        assert chi2 < 40, counts                         # 10 / 7 degrees of freedom: p(chi2 > 40) < 1e-4


def test_masked_actions_are_accepted_by_step():
    """every action the mask allows is feasible: stepping it never raises the QoT error / retry."""
    meta, d = load_traj("obs_nsfnet320_dense")
    env = make_env(meta, batch=64, load=900, margin=0.0)   # the mask ignores the margin (qrmsa.pyx:757), step() does not
    env.seed(5); env.reset()
    env.step_policy(300, record=False)
    rng = np.random.default_rng(0)
    for _ in range(20):
        _, mask = env.observe()
        acts = np.array([rng.choice(np.flatnonzero(m)) for m in mask], np.int32)
        rec = env.step(acts)
        assert not rec["retry"].any() and not (rec["flags"] & nat.F_QOT_ERROR).any()
        assert np.array_equal(rec["accepted"] == 0, acts == env.reject_action)


def test_vec_env_masked_random_policy_rollout():
    """VecEnv-shaped API: a masked random policy runs whole episodes; dones trigger the in-launch reset."""
    from optical_networking_gym.envs.vec_env import QRMSAVecEnv
    venv = QRMSAVecEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), num_envs=32, seed=3,
                       num_spectrum_resources=320, capacity=512, load=400, episode_length=60,
                       bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400))
    obs = venv.reset()
    assert obs.shape == (32, 368) and obs.dtype == np.float32
    rng = np.random.default_rng(1)
    finished = 0
    for t in range(150):
        masks = venv.action_masks()
        assert masks.shape == (32, 9601) and masks[:, -1].all()
        acts = [rng.choice(np.flatnonzero(m[:-1])) if m[:-1].any() and rng.random() < 0.95 else 9600 for m in masks]
        obs, rew, dones, infos = venv.step(acts)
        assert set(np.unique(rew)) <= {0.0, -6.0}
        assert (rew == -6.0).sum() == sum(a == 9600 for a in acts)
        for i in np.flatnonzero(dones):
            assert 0.0 <= infos[i]["episode"]["episode_service_blocking_rate"] <= 1.0
            finished += 1
        assert dones.sum() == (32 if (t + 1) % 59 == 0 else 0)      # every replica terminates after 59 steps
    assert finished == 64
    venv.close()
