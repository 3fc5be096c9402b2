"""Diagnostic: device vs oracle with defragmentation, one step per launch; prints the first divergence."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "optical-networking-gym_amd"))
import numpy as np
from common import load_traj, traj_requests, holder_for, golden_tables, jocn_modulations
from oracle_lib import OracleEnv
from optical_networking_gym.envs.batched import BatchedQRMSAEnv

tag = sys.argv[1] if len(sys.argv) > 1 else "traj_nsfnet320_defrag"
meta, d = load_traj(tag)
kw = dict(tables=golden_tables(meta["topology"]), modulations=jocn_modulations(), batch_size=1,
          num_spectrum_resources=meta["S"], episode_length=meta["episode_length"], load=meta["load"],
          mean_service_holding_time=meta["mean_holding"], bit_rate_selection=meta["bit_rate_selection"],
          bit_rates=tuple(meta["bit_rates"]), launch_power_dbm=meta["launch_power_dbm"],
          frequency_start=meta["frequency_start"], frequency_slot_bandwidth=meta["slot_bw"], margin=meta["margin"],
          capacity=1024, auto_reset=True, defragmentation=True, n_defrag_services=meta["n_defrag_services"])
env = BatchedQRMSAEnv(**kw)
env.set_requests(traj_requests(d))
o = OracleEnv(holder_for(meta, defragmentation=True, n_defrag_services=meta["n_defrag_services"]))
o.set_trace(traj_requests(d))
for _ in range(meta["initial_resets"]):
    env.reset(); o.reset()
for i in range(meta["n_steps"]):
    rec = env.step_policy(1)[0, 0]
    a, _, _ = o.policy_first_fit()
    rc, r = o.step(a)
    if r["terminated"]:
        o.reset()
    ga, gb = env.grid(0), o.grid()
    sa = np.sort(env.services(0), order="service_id"); sb = np.sort(o.services(), order="service_id")
    st, so = env.stats()[0], o.stats()
    bad = []
    if rec["action"] != r["action"]: bad.append(("action", rec["action"], r["action"]))
    if not np.array_equal(ga, gb): bad.append(("grid", np.argwhere(ga != gb)[:6].tolist()))
    if len(sa) != len(sb) or not all(np.array_equal(sa[f], sb[f]) for f in ("service_id", "path_id", "slot", "nslots")):
        bad.append(("services", len(sa), len(sb)))
    for f in ("episode_defrag_cycles", "episode_service_reallocations"):
        if st[f] != so[f]: bad.append((f, int(st[f]), int(so[f])))
    if bad:
        print("first divergence at step", i, bad)
        ids = set(sa["service_id"]) | set(sb["service_id"])
        for sid in sorted(ids):
            x = sa[sa["service_id"] == sid]; y = sb[sb["service_id"] == sid]
            if len(x) != len(y) or (len(x) and (x["slot"][0] != y["slot"][0] or x["path_id"][0] != y["path_id"][0])):
                print("  service", sid, "device", x[["path_id", "slot", "nslots", "modulation", "release_time", "osnr"]].tolist(),
                      "oracle", y[["path_id", "slot", "nslots", "modulation", "release_time", "osnr"]].tolist())
        break
else:
    print("no divergence over", meta["n_steps"], "steps")
