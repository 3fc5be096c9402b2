"""Single-environment compatibility view: the reference's `QRMSAEnv` surface on top of one device replica.

Mirrors `optical_networking_gym/envs/qrmsa.pyx` of the reference for the per-request hot path:
constructor kwargs (:206-237), `reset` (:427-504), `step` (:838-1065) and the helper API the heuristic plugins touch
(`get_number_slots` :1198-1205, `get_available_slots` :1482-1512, `_get_candidates` :515-541, `is_path_free`
:1248-1264, `encoded_decimal_to_array` :801-834, `get_available_blocks` :1513-1528, `_get_spectrum_slots` :1531-1541).

Every one of those calls is answered by the HIP kernels through the C ABI (include/ongym.h) on a B=1 environment:
this view does no slot-grid or GN arithmetic of its own, it only converts between the reference's Python objects
(`Service`, `Path`, graph attributes) and device records. It is the slow drop-in path for existing agents and plugin
heuristics; batch-scale use goes through `envs.batched.BatchedQRMSAEnv`.

`gen_observation=True` returns the device-computed observation vector and action mask (`ongym_observe`).
`measure_disruptions=True` counts disrupted services on device.
`defragmentation=True` runs the reference's defragment() on device and replays its moves onto the `Service` objects.
`bands` reproduces quirk Q9 (one slot per service); `file_name` writes the per-service CSV from the step records.
`reset(options={"only_episode_counters": True})` is the reference's counters-only reset (qrmsa.pyx:427-464).
A `k_paths` below the topology's k restricts routes, action space and device tables to the first `k_paths` routes of every
pair; a larger one raises (the reference would index past its route lists).
"""
from __future__ import annotations

import math
import os
from typing import Any, Optional

import numpy as np

from .. import _native as nat
from .._tables import StaticTables
from ..utils import rle
from .batched import BatchedQRMSAEnv, OngymError

try:  # gymnasium is optional (absent in the build image)
    import gymnasium as _gym
    _Discrete, _Box = _gym.spaces.Discrete, _gym.spaces.Box
except Exception:  # pragma: no cover - exercised where gymnasium is missing
    _gym = None

    class _Discrete:
        def __init__(self, n):
            self.n = int(n)
            self._rng = np.random.default_rng()

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class _Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Service:
    """Per-request record with the reference's field names (envs/qrmsa.pyx:29-116)."""

    __slots__ = ("service_id", "source", "source_id", "destination", "destination_id", "arrival_time",
                 "holding_time", "bit_rate", "path", "service_class", "initial_slot", "center_frequency", "bandwidth",
                 "number_slots", "core", "launch_power", "accepted", "blocked_due_to_resources",
                 "blocked_due_to_osnr", "OSNR", "ASE", "NLI", "current_modulation", "recalculate")

    def __init__(self, service_id, source, source_id, destination=None, destination_id=None, arrival_time=0.0,
                 holding_time=0.0, bit_rate=0.0, path=None, service_class=0, initial_slot=0, center_frequency=0,
                 bandwidth=0, number_slots=0, core=0, launch_power=0.0, accepted=False,
                 blocked_due_to_resources=True, blocked_due_to_osnr=True, OSNR=0.0, ASE=0.0, NLI=0.0,
                 current_modulation=None):
        self.service_id, self.source, self.source_id = service_id, source, source_id
        self.destination, self.destination_id = destination, destination_id
        self.arrival_time, self.holding_time, self.bit_rate = float(arrival_time), float(holding_time), float(bit_rate)
        self.path, self.service_class, self.initial_slot = path, service_class, initial_slot
        self.center_frequency, self.bandwidth, self.number_slots = center_frequency, bandwidth, number_slots
        self.core, self.launch_power, self.accepted = core, launch_power, accepted
        self.blocked_due_to_resources, self.blocked_due_to_osnr = blocked_due_to_resources, blocked_due_to_osnr
        self.OSNR, self.ASE, self.NLI = OSNR, ASE, NLI
        self.current_modulation = current_modulation
        self.recalculate = False

    def __repr__(self):
        return (f"Service(service_id={self.service_id}, source='{self.source}', destination='{self.destination}', "
                f"bit_rate={self.bit_rate}, path={self.path}, initial_slot={self.initial_slot}, "
                f"number_slots={self.number_slots}, accepted={self.accepted}, OSNR={self.OSNR})")


class _LazyViews(dict):
    """A `topology.graph` / edge-attribute dict whose state views (`available_slots`, `running_services`) are rebuilt from
    the device only when something reads them after the state changed: the reference keeps these entries current on every
    step (qrmsa.pyx:1291-1350); here a step that nobody inspects costs no host-side rebuild."""
    __slots__ = ("_owner", "_lazy")

    def __init__(self, data, owner, lazy):
        super().__init__(dict.items(data) if isinstance(data, dict) else data)   # no view rebuild while copying
        self._owner, self._lazy = owner, frozenset(lazy)

    def _fresh(self, key=None):
        if key is None or key in self._lazy:
            self._owner._ensure_views()

    def __getitem__(self, key):
        self._fresh(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._fresh(key)
        return dict.get(self, key, default)

    def items(self):
        self._fresh()
        return dict.items(self)

    def values(self):
        self._fresh()
        return dict.values(self)

    def copy(self):
        self._fresh()
        return dict(self)


class QRMSAEnv:
    def __init__(self, topology, num_spectrum_resources: int = 320, episode_length: int = 1000, load: float = 10.0,
                 mean_service_holding_time: float = 10800.0, bit_rate_selection: str = "continuous",
                 bit_rates: tuple = (10, 40, 100), bit_rate_probabilities=None, node_request_probabilities=None,
                 bit_rate_lower_bound: float = 25.0, bit_rate_higher_bound: float = 100.0,
                 launch_power_dbm: float = 0.0, bandwidth: float = 4e12, frequency_start: float = (3e8 / 1565e-9),
                 frequency_slot_bandwidth: float = 12.5e9, margin: float = 0.0, measure_disruptions: bool = False,
                 seed: object = None, allow_rejection: bool = True, reset: bool = True, channel_width: float = 12.5,
                 k_paths: int = 5, file_name: str = "", blocks_to_consider: int = 1, modulations_to_consider: int = 6,
                 defragmentation: bool = False, n_defrag_services: int = 0, gen_observation: bool = True,
                 bands: object = None, device: int = 0, capacity: int = 1024, sync_views: bool = True,
                 requests: Optional[np.ndarray] = None):
        self.gen_observation = bool(gen_observation)
        self.defragmentation, self.n_defrag_services = bool(defragmentation), int(n_defrag_services)
        self.measure_disruptions = bool(measure_disruptions)
        if seed is not None and not isinstance(seed, (int, np.integer)):
            raise ValueError("Seed must be an integer.")
        self.topology = topology
        self.k_shortest_paths = topology.graph["ksp"]          # KeyError 'ksp' on a bare graph, like the reference
        topo_k = int(topology.graph.get("k_paths") or max((len(r) for r in self.k_shortest_paths.values()), default=0))
        if int(k_paths) > topo_k:
            raise ValueError(f"k_paths={k_paths} but the topology was built with {topo_k} routes per node pair")
        if int(k_paths) < topo_k:
            # the env's k_paths rules (action space, reject action, observation, qrmsa.pyx:319-321, 380, 696-700): plugins and
            # the device then see only the first k_paths routes of every pair
            self.k_shortest_paths = {pair: routes[:int(k_paths)] for pair, routes in self.k_shortest_paths.items()}
        self.modulations = topology.graph.get("modulations", [])
        self.num_spectrum_resources = int(num_spectrum_resources)
        self.episode_length = int(episode_length)
        self.load, self.mean_service_holding_time = float(load), float(mean_service_holding_time)
        self.bit_rate_selection, self.bit_rates = bit_rate_selection, tuple(bit_rates)
        self.channel_width, self.k_paths = float(channel_width), int(k_paths)
        self.launch_power_dbm = float(launch_power_dbm)
        self.launch_power = 10 ** ((self.launch_power_dbm - 30) / 10)
        self.frequency_start, self.frequency_slot_bandwidth = float(frequency_start), float(frequency_slot_bandwidth)
        self.frequency_end = self.frequency_start + self.frequency_slot_bandwidth * self.num_spectrum_resources
        assert math.isclose(self.frequency_end - self.frequency_start, bandwidth, rel_tol=1e-5)
        self.margin = float(margin)
        self.max_modulation_idx = len(self.modulations) - 1
        self.modulations_to_consider = min(modulations_to_consider, len(self.modulations))
        self.allow_rejection = allow_rejection
        # multiband quirk Q9 (qrmsa.pyx:417-425, 1198-1205): with bands the slot count is computed with the C band's
        # channel width in Hz, i.e. every service needs one slot
        self.bands = list(bands) if bands else []
        self.current_band = self.bands[1] if self.bands else None
        self._slot_width = ((self.current_band.freq_end - self.current_band.freq_start) * 1e12 / self.current_band.num_slots
                            if self.current_band is not None else self.channel_width)
        self.blocks_to_consider = blocks_to_consider
        self.input_seed = int(seed) % (2 ** 31) if seed is not None else int(np.random.SeedSequence().generate_state(1)[0] % (2 ** 31))
        self.action_space = _Discrete(self.k_paths * self.modulations_to_consider * self.num_spectrum_resources + 1)
        self.observation_space = _Box(low=-5, high=5, shape=(1 + 2 + self.k_paths + self.k_paths * self.modulations_to_consider * 12,),
                                      dtype=np.float32)
        self.reject_action = self.action_space.n - 1 if allow_rejection else 0
        self._tables = StaticTables.from_topology(topology).truncated(self.k_paths)
        self._paths_by_id = {}
        for routes in self.k_shortest_paths.values():
            for p in routes:
                self._paths_by_id[int(p.id)] = p
        self._nodes = list(topology.graph["node_indices"])
        dev_kw = dict(
            tables=self._tables, modulations=self.modulations, modulations_to_consider=modulations_to_consider,
            batch_size=1, capacity=capacity, auto_reset=False, device=device,
            num_spectrum_resources=num_spectrum_resources, episode_length=episode_length, load=load,
            mean_service_holding_time=mean_service_holding_time, bit_rate_selection=bit_rate_selection,
            bit_rates=bit_rates, bit_rate_probabilities=bit_rate_probabilities,
            node_request_probabilities=node_request_probabilities, bit_rate_lower_bound=bit_rate_lower_bound,
            bit_rate_higher_bound=bit_rate_higher_bound, launch_power_dbm=launch_power_dbm,
            frequency_start=frequency_start, frequency_slot_bandwidth=frequency_slot_bandwidth, margin=margin,
            channel_width=self.channel_width, nslots_channel_width=(self._slot_width if self.current_band is not None else 0.0),
            measure_disruptions=measure_disruptions,
            defragmentation=defragmentation, n_defrag_services=n_defrag_services)
        # service ids are kept on device so that calculate_osnr's skip-by-service-id (core/osnr.pyx:65) is exact also after a
        # counters-only reset; the id-tracking kernels need uniform attenuation (per-link attenuation: no ids, and the
        # counters-only reset is then refused)
        try:
            self._dev = BatchedQRMSAEnv(track_service_ids=True, **dev_kw)
            self._tracks_ids = True
        except OngymError as exc:
            if "uniform attenuation" not in str(exc):
                raise
            self._dev = BatchedQRMSAEnv(**dev_kw)
            self._tracks_ids = False
        if requests is not None:
            self._dev.set_requests(requests)           # trace replay (parity tests)
        else:
            # the reference's traffic RNG is unseeded (quirk Q2); here `seed` selects the device stream
            self._dev.seed(self.input_seed)
        self._sync_views = bool(sync_views)
        self._state_version, self._views_version = 1, 0
        if self._sync_views:   # the state views are rebuilt on first read after a change (see _LazyViews)
            if not isinstance(topology.graph, _LazyViews) or topology.graph._owner is not self:
                topology.graph = _LazyViews(topology.graph, self, ("available_slots", "running_services"))
            for u, v in topology.edges():
                lv = _LazyViews(topology._adj[u][v], self, ("running_services",))
                topology._adj[u][v] = lv
                topology._adj[v][u] = lv
        self.file_stats = None
        if file_name != "":   # per-service CSV of qrmsa.pyx:387-406, same name pattern and header
            final_name = "_".join([file_name, str(topology.graph["name"]), str(self.launch_power_dbm), str(self.load),
                                   str(seed) + ".csv"])
            directory = os.path.dirname(final_name)
            if directory and not os.path.exists(directory):
                os.makedirs(directory, exist_ok=True)
            self.final_file_name = final_name
            self.file_stats = open(final_name, "wt", encoding="UTF-8")
            self.file_stats.write("# Service stats file from simulator\n")
            self.file_stats.write("id,source,destination,bit_rate,path_k,path_length,modulation,min_osnr,osnr,ase,nli,"
                                  "disrupted_services,active_services\n")
        self._disrupted_seen = 0
        self._sticky_policy, self._policy_cache = -1, None     # see policy_action / step
        self._modulation_keys = ["modulation_{}".format(str(float(m.spectral_efficiency))) for m in self.modulations]
        self._action_buf = np.zeros(1, np.int32)
        self.current_service: Optional[Service] = None
        self.current_time = 0.0
        self._last_stats = None
        if reset:
            self.reset()

    # ---- views ---------------------------------------------------------------------------------------------------
    def _pull_request(self, q=None, st=None):
        if q is None:
            q = self._dev.request(0)
            st = self._dev.stats()[0]
        self._last_stats = st
        self.current_time = float(st["current_time"])
        src, dst = int(q["source"]), int(q["destination"])
        self.current_service = Service(
            service_id=int(st["episode_services_processed"]) - 1, source=self._nodes[src], source_id=src,
            destination=self._nodes[dst], destination_id=str(dst), arrival_time=float(q["arrival_time"]),
            holding_time=float(q["holding_time"]), bit_rate=float(q["bit_rate"]))

    def _refresh_views(self):
        """The device state changed: the views are stale.  With defragmentation the rebuild also replays moved services
        onto the listed Service objects, which readers reach without touching a view, so it stays eager."""
        self._state_version += 1
        if self.defragmentation:
            self._ensure_views()

    def _ensure_views(self):
        if not self._sync_views or self._views_version == self._state_version:
            return
        self._views_version = self._state_version      # first: the rebuild itself reads and writes the dicts
        g = self.topology.graph
        g["available_slots"] = self._dev.grid(0)
        running = []
        for u, v in self.topology.edges():
            self.topology[u][v]["running_services"] = []
        for rec in self._dev.services(0):
            path = self._paths_by_id[int(rec["path_id"])]
            mod = self.modulations[int(rec["modulation"])]
            n, s = int(rec["nslots"]), int(rec["slot"])
            if self.defragmentation:   # defragment() moves services and rewrites their OSNR (qrmsa.pyx:1590-1632)
                listed = self._accepted_by_id.get(int(rec["service_id"]))
                if listed is not None and listed.initial_slot != s:
                    listed.initial_slot = s
                    listed.center_frequency = (self.frequency_start + self.frequency_slot_bandwidth * s
                                               + self.frequency_slot_bandwidth * (n / 2.0))
                    listed.OSNR = float(rec["osnr"])
            svc = Service(service_id=int(rec["service_id"]), source=path.node_list[0], source_id=self._nodes.index(path.node_list[0]),
                          destination=path.node_list[-1], path=path, initial_slot=s, number_slots=n,
                          center_frequency=self.frequency_start + self.frequency_slot_bandwidth * s
                          + self.frequency_slot_bandwidth * (n / 2.0),
                          bandwidth=self.frequency_slot_bandwidth * n, launch_power=self.launch_power, accepted=True,
                          current_modulation=mod)
            running.append(svc)
            for link in path.links:
                self.topology[link.node1][link.node2]["running_services"].append(svc)
        g["running_services"] = running

    # ---- gym surface ---------------------------------------------------------------------------------------------------
    def _blank_observation(self):
        if self.gen_observation:   # observation() + action mask of the CURRENT request, computed on device (qrmsa.pyx:583-781)
            obs, mask = self._dev.observe()
            # observation() moved the codec's format window (get_max_modulation_index, qrmsa.pyx:543-581, 680)
            self.max_modulation_idx = int(self._dev.stats()[0]["max_modulation_idx"])
            return obs[0], {"mask": mask[0]}
        # gen_observation=False: zeros, including the reject slot of the mask (qrmsa.pyx:584-587)
        return (np.zeros(self.observation_space.shape, np.float32),
                {"mask": np.zeros(self.action_space.n, np.uint8)})

    def reset(self, seed=None, options=None):
        if options is not None and options.get("only_episode_counters"):
            # qrmsa.pyx:427-464: episode counters and histograms restart, `self._events = []` drops the departure heap (the
            # running services stay for good), nothing else changes and no request is drawn; returns (observation, {})
            if not self._tracks_ids:
                raise NotImplementedError("only_episode_counters needs service ids on device (uniform attenuation)")
            self._dev.reset_episode_counters()
            self._policy_cache = None          # service ids restart: the GN model's "self" skip (core/osnr.pyx:65) may decide otherwise
            self.max_modulation_idx = len(self.modulations) - 1
            self._last_stats = self._dev.stats()[0]
            obs, _ = self._blank_observation()
            return obs, {}
        self._dev.reset()
        self.topology.graph["services"] = []
        self._accepted_by_id = {}
        for u, v in self.topology.edges():     # qrmsa.pyx:475-495
            link = self.topology[u][v]
            link["utilization"] = link["last_update"] = link["external_fragmentation"] = link["compactness"] = 0.0
        self.max_modulation_idx = len(self.modulations) - 1
        self._pull_request()
        self._refresh_views()
        self._ensure_views()           # once per reset: the view keys exist from here on
        obs, mask = self._blank_observation()
        return obs, dict(mask)

    def step(self, action: int):
        cur = self.current_service
        running_before = len(self._dev.services(0)) if self.file_stats is not None else 0
        # one call: the step, the next request, the statistics and - for the fused heuristic the caller used last - the next
        # decision (graph_load.py:157-164 calls heuristic(env) and env.step(action) in turn: one synchronisation per iteration)
        self._policy_cache = None
        self._action_buf[0] = int(action)
        recs, reqs, sts, nacts, nflags = self._dev.step_bundle(self._action_buf, self._sticky_policy)
        rec = recs[0]
        obs, mask = (None, None) if (rec["flags"] & (nat.F_QOT_ERROR | nat.F_NO_REQUEST)) else self._blank_observation()
        if rec["flags"] & nat.F_QOT_ERROR:
            route, mod_idx, slot = self.encoded_decimal_to_array(int(action))
            modulation = self.modulations[mod_idx]
            raise ValueError(f"Osnr {rec['osnr']} is not enough for service {cur.service_id} with modulation "
                             f"{modulation}, and osnr_req {modulation.minimum_osnr + self.margin}.")
        if rec["flags"] & nat.F_NO_REQUEST:
            raise OngymError("request trace exhausted")
        if rec["retry"]:   # quirk Q5 (qrmsa.pyx:886-897): penalty, same request stays current
            cur.blocked_due_to_resources, cur.accepted = True, False
            info = {"blocked_due_to_resources": 1, "blocked_due_to_osnr": 0, "rejected": 1}
            info.update(mask)
            return obs, float(rec["reward"]), False, False, info
        cur.blocked_due_to_resources = cur.blocked_due_to_osnr = False
        cur.accepted = bool(rec["accepted"])
        osnr_req = 0.0
        if int(action) != self.action_space.n - 1:
            osnr_req = self.modulations[self.encoded_decimal_to_array(int(action))[1]].minimum_osnr + self.margin
        if cur.accepted:
            n, s = int(rec["nslots"]), int(rec["slot"])
            cur.path = self.k_shortest_paths[cur.source, cur.destination][int(rec["route"])]
            cur.initial_slot, cur.number_slots = s, n
            cur.center_frequency = self.frequency_start + self.frequency_slot_bandwidth * s + self.frequency_slot_bandwidth * (n / 2.0)
            cur.bandwidth, cur.launch_power = self.frequency_slot_bandwidth * n, self.launch_power
            cur.OSNR, cur.ASE, cur.NLI = float(rec["osnr"]), float(rec["ase"]), float(rec["nli"])
            cur.current_modulation = self.modulations[int(rec["modulation"])]
        else:
            cur.path, cur.initial_slot, cur.number_slots = None, -1, 0
            cur.OSNR = cur.ASE = cur.NLI = 0.0
        self.topology.graph["services"].append(cur)
        if self.defragmentation:
            if cur.accepted:
                self._accepted_by_id[cur.service_id] = cur
            # defragment() ran inside this step's _next_service: replay its moves onto the Service objects (a moved
            # service may already have departed again, so the running-services view alone would miss it)
            moves, total = self._dev.moves(0)
            for mv in moves:
                listed = self._accepted_by_id.get(int(mv["service_id"]))
                if listed is not None:
                    listed.initial_slot = int(mv["slot"])
                    listed.center_frequency = (self.frequency_start + self.frequency_slot_bandwidth * listed.initial_slot
                                               + self.frequency_slot_bandwidth * (listed.number_slots / 2.0))
                    listed.launch_power = self.launch_power
                    listed.OSNR, listed.ASE, listed.NLI = float(mv["osnr"]), float(mv["ase"]), float(mv["nli"])
        self._pull_request(reqs[0], sts[0])
        self._refresh_views()
        if nacts is not None and not rec["retry"]:
            self._policy_cache = (self._sticky_policy, self._state_version, int(nacts[0]), int(nflags[0]))
        st = self._last_stats
        terminated = bool(rec["terminated"])
        if self.file_stats is not None:   # qrmsa.pyx:967-990
            newly = int(st["disrupted_services"]) - self._disrupted_seen if int(st["disrupted_services"]) >= self._disrupted_seen else int(st["disrupted_services"])
            self._disrupted_seen = int(st["disrupted_services"])
            line = "{},{},{},{},".format(cur.service_id, cur.source_id, cur.destination_id, cur.bit_rate)
            if cur.accepted:
                line += "{},{},{},{},{},{},{},{},{}".format(
                    cur.path.k, cur.path.length, cur.current_modulation.spectral_efficiency,
                    cur.current_modulation.minimum_osnr, cur.OSNR, cur.ASE, cur.NLI, newly, running_before + 1)
            else:
                line += "-1,-1,-1,-1,-1,-1,-1,-1,-1"
            self.file_stats.write(line + "\n")
            self.file_stats.flush()
        # info of qrmsa.pyx:996-1050 is computed BEFORE the next request is drawn: undo that draw's increments
        sp, sa = int(st["services_processed"]) - 1, int(st["services_accepted"])
        ep, ea = int(st["episode_services_processed"]) - 1, int(st["episode_services_accepted"])
        nxt = float(self.current_service.bit_rate)
        brq, brp = float(st["bit_rate_requested"]) - nxt, float(st["bit_rate_provisioned"])
        ebrq, ebrp = float(st["episode_bit_rate_requested"]) - nxt, float(st["episode_bit_rate_provisioned"])
        info = {
            "episode_services_accepted": ea,
            "service_blocking_rate": (sp - sa) / sp if sp > 0 else 0.0,
            "episode_service_blocking_rate": (ep - ea) / ep if ep > 0 else 0.0,
            "bit_rate_blocking_rate": (brq - brp) / brq if brq > 0 else 0.0,
            "episode_bit_rate_blocking_rate": (ebrq - ebrp) / ebrq if ebrq > 0 else 0.0,
            # :1035-1041 — the global ratio is a true division, the episode one divides two C ints (always 0 unless
            # every accepted service was disrupted)
            "disrupted_services": (float(st["disrupted_services"]) / sa) if (st["disrupted_services"] > 0 and sa > 0) else 0.0,
            "episode_disrupted_services": float(int(st["episode_disrupted_services"]) // ea) if (st["episode_disrupted_services"] > 0 and ea > 0) else 0.0,
            "osnr": float(rec["osnr"]), "osnr_req": float(osnr_req),
            "chosen_path_index": int(rec["route"]), "chosen_slot": int(rec["slot"]),
            "episode_defrag_cicles": int(st["step_defrag_cycles"]),
            "episode_service_realocations": int(st["step_service_reallocations"]),
        }
        if terminated:   # the device snapshots the exact fp64 rates at the terminal step
            info["service_blocking_rate"] = float(st["last_service_blocking_rate"])
            info["episode_service_blocking_rate"] = float(st["last_episode_service_blocking_rate"])
            info["bit_rate_blocking_rate"] = float(st["last_bit_rate_blocking_rate"])
            info["episode_bit_rate_blocking_rate"] = float(st["last_episode_bit_rate_blocking_rate"])
        hist = st["episode_modulation_hist"]
        for m, key in enumerate(self._modulation_keys):
            info[key] = int(hist[m])
        if terminated:
            info["blocked_due_to_resources"] = 0   # always 0 in the reference (quirk Q6)
            info["blocked_due_to_osnr"] = 0
            info["rejected"] = int(st["last_rejected"])
        info.update(mask)
        return obs, float(rec["reward"]), terminated, False, info

    # ---- plugin API --------------------------------------------------------------------------------------------------
    def get_number_slots(self, service, modulation) -> int:
        # integer ceil of bit_rate / (SE * 12.5): answered by the same table the kernels use is not exposed per call;
        # this is pure request arithmetic (no state), identical to qrmsa.pyx:1198-1205 with bands unset
        return int(math.ceil(float(np.float32(service.bit_rate)) / (modulation.spectral_efficiency * self._slot_width)))

    def get_available_slots(self, path) -> np.ndarray:
        return self._dev.available_slots(0, int(path.id))

    def _get_candidates(self, available_slots, num_slots_required, total_slots):
        row = np.asarray(available_slots)[:total_slots]
        return self._dev.candidates(row, int(num_slots_required))

    def is_path_free(self, path, initial_slot: int, number_slots: int) -> bool:
        if initial_slot < 0 or initial_slot >= self.num_spectrum_resources:
            return False
        return self._dev.is_path_free(0, int(path.id), int(initial_slot), int(number_slots))

    def calculate_osnr(self, service) -> tuple:
        g = self._dev.gsnr(0, int(service.path.id), int(service.initial_slot), int(service.number_slots))
        return float(g[0]), float(g[1]), float(g[2])

    def calculate_osnr_many(self, candidates) -> np.ndarray:
        """`calculate_osnr` for a list of `(path, initial_slot, number_slots)` candidates in one device launch.
        Returns float64 [len][3] = (gsnr, ase, nli) dB.  Not in the reference: plugins that score every feasible start
        (heuristics.py:272-328, 330-416, 647-749) use it instead of one query per candidate."""
        tri = np.array([(int(p.id), int(s), int(n)) for p, s, n in candidates], np.int32).reshape(-1, 3)
        return self._dev.gsnr_many(0, tri)

    def _allowed_modulations(self):
        M = self.modulations_to_consider
        if self.max_modulation_idx > 1:
            return list(range(self.max_modulation_idx, self.max_modulation_idx - M, -1))
        return list(reversed(range(M)))

    def encoded_decimal_to_array(self, decimal: int, max_values=None):
        if max_values is None:
            max_values = [self.k_paths, self.modulations_to_consider, self.num_spectrum_resources]
        digits = []
        for radix in reversed(max_values):
            digits.insert(0, decimal % radix)
            decimal //= radix
        digits[1] = self._allowed_modulations()[digits[1]]
        return digits

    decimal_to_array = encoded_decimal_to_array

    def get_available_blocks(self, path: int, slots: int, j):
        avail = self.get_available_slots(self.k_shortest_paths[self.current_service.source,
                                                               self.current_service.destination][path])
        starts, values, lengths = rle(avail)
        keep = np.intersect1d(np.where(values == 1)[0], np.where(lengths >= slots)[0])[:j]
        return starts[keep], lengths[keep]

    def _get_spectrum_slots(self, path: int):
        grid = self._dev.grid(0)
        route = self.k_shortest_paths[self.current_service.source, self.current_service.destination][path]
        return [grid[self.topology[l.node1][l.node2]["index"], :] for l in route.links]

    def _get_network_compactness(self) -> float:
        """Spectrum compactness of the whole network, qrmsa.pyx:1150-1186: (occupied span summed over links / slot-hops of the
        running services) x (links / free blocks inside the occupied spans); 1.0 when no link has a free block inside its
        occupied span.  Parity unpinned: the compiled reference segfaults in this method (its `rle` is handed a row of the wrong
        element type), so no output of it exists to compare with - this is the arithmetic of the source text on the device-
        backed views.  (An empty network returns 1.0 here; the source text would divide by zero slot-hops only if a link held
        two used blocks without any running service, which cannot happen.)"""
        slot_hops = sum(s.number_slots * s.path.hops for s in self.topology.graph["running_services"])
        grid = np.asarray(self.topology.graph["available_slots"])
        occupied = unused_blocks = 0.0
        for n1, n2 in self.topology.edges():
            row = grid[self.topology[n1][n2]["index"], :]
            starts, values, lengths = rle(row)
            used = np.flatnonzero(values == 0)
            if len(used) > 1:
                lo = int(starts[used[0]])
                hi = int(starts[used[-1]] + lengths[used[-1]])
                occupied += hi - lo
                _, inner_values, _ = rle(row[lo:hi])
                unused_blocks += float(np.sum(inner_values))        # one per run of value 1 inside the span
        if unused_blocks > 0:
            return (occupied / slot_hops) * (self.topology.number_of_edges() / unused_blocks)
        return 1.0

    def _update_link_stats(self, node1, node2) -> None:
        """Time-weighted utilisation / external fragmentation / compactness of one link (qrmsa.pyx:1353-1480), written
        to the edge attributes like the reference.  Host-side arithmetic on the device's grid view; nothing in the
        reference calls it, it is kept for scripts that do.  Its naming quirks are kept: the "total unused slots" of
        the fragmentation term is the number of USED slots, the compactness divisor counts used BLOCKS."""
        link = self.topology[node1][node2]
        last_update = link["last_update"]
        last_fragmentation = link.get("external_fragmentation", 0.0)
        last_compactness = link.get("compactness", 0.0)
        now = self.current_time
        time_diff = now - last_update
        row = np.asarray(self.topology.graph["available_slots"][link["index"], :], dtype=np.int32)
        free = int(row.sum())
        if now > 0:
            cur_util = (self.num_spectrum_resources - free) / self.num_spectrum_resources
            link["utilization"] = ((link["utilization"] * last_update) + (cur_util * time_diff)) / now
        starts, values, lengths = rle(row)
        free_runs = np.flatnonzero(values == 1)
        if len(free_runs) > 1 and free_runs.tolist() != [0, len(values) - 1]:
            max_empty = int(lengths[free_runs].max())
        else:
            max_empty = 0
        if free > 0:
            with np.errstate(divide="ignore", invalid="ignore"):   # an idle link divides 0 by 0 here, like the reference
                cur_fragmentation = float(1.0 - np.float64(max_empty) / np.float64(row.shape[0] - free))
        else:
            cur_fragmentation = 1.0
        used_runs = np.flatnonzero(values == 0)
        cur_compactness = 1.0
        if len(used_runs) > 1:
            lo = int(starts[used_runs[0]])
            hi = int(starts[used_runs[-1]] + lengths[used_runs[-1]])
            _, inner_values, _ = rle(row[lo:hi])
            used_blocks = float(np.sum(1 - inner_values))
            used_slots = float(np.sum(1 - row))
            if used_blocks > 0 and used_slots > 0:
                cur_compactness = ((hi - lo) / used_slots) * (1.0 / used_blocks)
        with np.errstate(divide="ignore", invalid="ignore"):
            link["external_fragmentation"] = float(
                (np.float64(last_fragmentation * last_update) + np.float64(cur_fragmentation * time_diff)) / np.float64(now))
            link["compactness"] = float(
                (np.float64(last_compactness * last_update) + np.float64(cur_compactness * time_diff)) / np.float64(now))
        link["last_update"] = now

    def policy_action(self, policy: int = nat.POLICY_FIRST_FIT):
        """A fused device policy evaluated on the current request: (action, blocked_resources, blocked_osnr)."""
        c = self._policy_cache
        if c is not None and c[0] == policy and c[1] == self._state_version:     # decided by the last step's bundle
            a0, f0 = c[2], c[3]
        else:
            a, f = self._dev.policy_actions(policy)
            a0, f0 = int(a[0]), int(f[0])
        self._sticky_policy = int(policy)      # the next step() asks the device for this policy's next decision in the same call
        return a0, bool(f0 & nat.F_BLOCKED_RESOURCES), bool(f0 & nat.F_BLOCKED_OSNR)

    def first_fit_action(self):
        """heuristic_shortest_available_path_first_fit_best_modulation (heuristics.py:923-966) on device."""
        return self.policy_action(nat.POLICY_FIRST_FIT)

    def close(self):
        if self.file_stats is not None:
            self.file_stats.close()
            self.file_stats = None
        self._dev.close()
