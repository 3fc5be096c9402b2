"""Policy plugins with the reference's calling convention: `f(env) -> (action, blocked_resources, blocked_osnr)`.

Reference: optical_networking_gym/heuristics/heuristics.py — `get_qrmsa_env` (:15-33), `get_action_index` (:36-54),
`heuristic_shortest_available_path_first_fit_best_modulation` (:923-966), `heuristic_highest_snr` (:272-328).

* First fit, highest SNR and load-balancing-best-modulation are answered by the policies fused on device
  (`ongym_policy_actions`).
* `heuristic_shortest_available_path_first_fit_best_modulation_plugin` is the same policy written against the plugin API
  only (k_shortest_paths / get_number_slots / get_available_slots / _get_candidates / calculate_osnr): it exists to show
  — and test — that plugins written for the reference run unchanged on the compatibility view.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from .. import _native as _nat
from ..core.osnr import calculate_osnr
from ..utils import fragmentation_route_cuts, fragmentation_route_rss, link_shannon_entropy_, rle
from ..envs.qrmsa import QRMSAEnv


def get_qrmsa_env(env) -> QRMSAEnv:
    """Unwrap `.env` chains until the base QRMSAEnv is found."""
    while not isinstance(env, QRMSAEnv):
        if not hasattr(env, "env"):
            raise ValueError("QRMSAEnv was not found in the wrapper chain of the environment.")
        env = env.env
    return env


def get_action_index(env: QRMSAEnv, path_index: int, modulation_index: int, initial_slot: int) -> int:
    relative = env.max_modulation_idx - modulation_index
    return (path_index * env.modulations_to_consider + relative) * env.num_spectrum_resources + initial_slot


def heuristic_shortest_available_path_first_fit_best_modulation(env):
    """First path (shortest first) x best modulation (most efficient first) x lowest feasible slot whose GSNR clears
    the threshold + margin; else the reject action. Evaluated by the fused device policy."""
    return get_qrmsa_env(env).first_fit_action()


def _stage_candidate(sim_env, service, path, modulation, slot, slots):
    service.path, service.initial_slot, service.number_slots = path, slot, slots
    service.current_modulation = modulation
    service.center_frequency = (sim_env.frequency_start + sim_env.frequency_slot_bandwidth * slot
                                + sim_env.frequency_slot_bandwidth * (slots / 2))
    service.bandwidth = sim_env.frequency_slot_bandwidth * slots
    service.launch_power = sim_env.launch_power


def heuristic_shortest_available_path_first_fit_best_modulation_plugin(env):
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    no_slots = low_osnr = False
    for path_idx, path in enumerate(sim_env.k_shortest_paths[service.source, service.destination]):
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            if slots <= 0:
                continue
            starts = sim_env._get_candidates(sim_env.get_available_slots(path), slots, sim_env.num_spectrum_resources)
            if not starts:
                no_slots = True
                continue
            _stage_candidate(sim_env, service, path, modulation, starts[0], slots)
            osnr, _, _ = calculate_osnr(sim_env, service)
            if osnr >= modulation.minimum_osnr + sim_env.margin:
                return get_action_index(sim_env, path_idx, modulation_idx, starts[0]), False, False
            low_osnr, no_slots = True, False
    return env.action_space.n - 1, no_slots, low_osnr


def heuristic_highest_snr(env):
    """Every valid start of every (path, modulation) pair; highest GSNR above threshold wins (reference :272-328).
    Evaluated by the fused device policy (`ONGYM_POLICY_HIGHEST_SNR`)."""
    return get_qrmsa_env(env).policy_action(_nat.POLICY_HIGHEST_SNR)


def load_balancing_best_modulation(env):
    """Least-loaded of the k paths, best modulation, first fit (reference :547-627). Fused device policy."""
    return get_qrmsa_env(env).policy_action(_nat.POLICY_LOAD_BALANCING)


def shortest_available_path_first_fit_best_modulation(mask: np.ndarray) -> Optional[int]:
    """Mask-based first fit (reference :419-422): the first allowed action index."""
    return int(np.flatnonzero(np.asarray(mask) == 1)[0])


def rnd(mask: np.ndarray) -> Optional[int]:
    """Uniformly random allowed action (reference :424-428)."""
    return int(np.random.choice(np.flatnonzero(np.asarray(mask) == 1)))


# ----------------------------------------------------------------------------------------------------------------------
# The remaining reference policies, written against the plugin API.  They are pure functions of the env state, so the
# candidates they score are evaluated on the device in ONE launch (`calculate_osnr_many`) where the reference loops.
# ----------------------------------------------------------------------------------------------------------------------
def _routes(sim_env):
    service = sim_env.current_service
    return sim_env.k_shortest_paths[service.source, service.destination]


def _passes(sim_env, modulation, osnr) -> bool:
    return osnr >= modulation.minimum_osnr + sim_env.margin


def _first_fit_options(sim_env, routes=None):
    """(path_idx, path, modulation_idx, modulation, slots, avail, starts) for every (route, modulation) in the order the
    reference nests them: routes outer, modulations from `max_modulation_idx` down to 0 inner (heuristics.py:929-940)."""
    service = sim_env.current_service
    for path_idx, path in (routes if routes is not None else enumerate(_routes(sim_env))):
        avail = sim_env.get_available_slots(path)
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            if slots <= 0:
                continue
            starts = sim_env._get_candidates(avail, slots, sim_env.num_spectrum_resources)
            yield path_idx, path, modulation_idx, modulation, slots, avail, starts


def _osnr_at(sim_env, path, modulation, slot, slots) -> float:
    service = sim_env.current_service
    _stage_candidate(sim_env, service, path, modulation, slot, slots)
    return calculate_osnr(sim_env, service)[0]


def shortest_available_path_lowest_spectrum_best_modulation_plugin(env):
    """Reference :431-490 — the first-fit walk; only the flag rule differs (an OSNR failure anywhere clears the
    resources flag at the end)."""
    sim_env = get_qrmsa_env(env)
    no_slots = low_osnr = False
    for path_idx, path, modulation_idx, modulation, slots, _, starts in _first_fit_options(sim_env):
        if not starts:
            no_slots = True
            continue
        if _passes(sim_env, modulation, _osnr_at(sim_env, path, modulation, starts[0], slots)):
            return get_action_index(sim_env, path_idx, modulation_idx, starts[0]), False, False
        low_osnr = True
    return env.action_space.n - 1, (no_slots and not low_osnr), low_osnr


def best_modulation_load_balancing_plugin(env):
    """Reference :491-545 — modulations outer (ALL of them, most efficient first), routes inner; the first free run of
    at least slots+1 (no exception at the end of the spectrum) whose GSNR passes.  Never reports a blocking cause."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    rows = [sim_env.get_available_slots(path) for path in _routes(sim_env)]
    for modulation_idx in range(len(sim_env.modulations) - 1, -1, -1):
        modulation = sim_env.modulations[modulation_idx]
        slots = sim_env.get_number_slots(service, modulation)
        for path_idx, path in enumerate(_routes(sim_env)):
            starts, values, lengths = rle(rows[path_idx])
            fit = np.flatnonzero((values == 1) & (lengths >= slots + 1))
            if fit.size == 0:
                continue
            slot = int(starts[fit[0]])
            if _passes(sim_env, modulation, _osnr_at(sim_env, path, modulation, slot, slots)):
                return get_action_index(sim_env, path_idx, modulation_idx, slot), False, False
    return sim_env.reject_action, False, False


def heuristic_load_balancing_first_fit_plugin(env):
    """Reference :202-269 — routes sorted by (occupied fraction, route index), then first fit; (reject, True, False)."""
    sim_env = get_qrmsa_env(env)
    ranked = []
    for path_idx, path in enumerate(_routes(sim_env)):
        avail = sim_env.get_available_slots(path)
        ranked.append((np.sum(avail == 0) / len(avail) if len(avail) > 0 else 1.0, path_idx, path))
    ranked.sort(key=lambda t: (t[0], t[1]))
    for path_idx, path, modulation_idx, modulation, slots, _, starts in _first_fit_options(
            sim_env, [(idx, path) for _, idx, path in ranked]):
        if starts and _passes(sim_env, modulation, _osnr_at(sim_env, path, modulation, starts[0], slots)):
            return get_action_index(sim_env, path_idx, modulation_idx, starts[0]), False, False
    return sim_env.action_space.n - 1, True, False


def heuristic_lowest_fragmentation_plugin(env):
    """Reference :330-416, written against the plugin API (one batched GSNR launch per decision).  Quirks kept: the request is sized slots+1 (and evaluated by the GN model at that width); the
    trial allocation paints 1 (= free) over slots that are free already, so the fragmentation score (0.33*entropy +
    0.33*cuts + 0.34*rss over the ZERO-runs of the route's link rows) is one number per route; strict `<` keeps the first
    candidate of the lowest-score route whose GSNR passes."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    S = sim_env.num_spectrum_resources
    options, cands = [], []
    no_slots = low_osnr = False
    for path_idx, path in enumerate(_routes(sim_env)):
        rows = np.stack([np.asarray(r) for r in sim_env._get_spectrum_slots(path_idx)], axis=0)
        avail = sim_env.get_available_slots(path)
        score = None
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation) + 1
            if slots <= 0:
                continue
            starts = sim_env._get_candidates(avail, slots, S)
            if not starts:
                no_slots = True
                continue
            if score is None:
                lists = [row.tolist() for row in rows]
                entropy = [link_shannon_entropy_(row) for row in lists]
                score = (0.33 * (sum(entropy) / len(entropy) if entropy else 0.0)
                         + 0.33 * fragmentation_route_cuts(lists) + 0.34 * fragmentation_route_rss(lists))
            for start in starts:
                options.append((score, path_idx, modulation_idx, modulation, start))
                cands.append((path, start, slots))
    best, best_score = None, math.inf
    if cands:
        osnr = sim_env.calculate_osnr_many(cands)[:, 0]
        for (score, path_idx, modulation_idx, modulation, start), value in zip(options, osnr):
            if not _passes(sim_env, modulation, value):
                low_osnr = True
            elif score < best_score:
                best, best_score = get_action_index(sim_env, path_idx, modulation_idx, start), score
    if best is not None:
        return best, False, False
    return env.action_space.n - 1, (no_slots and not low_osnr), low_osnr


def _window_counts(row: np.ndarray, width: int) -> np.ndarray:
    """cum[i] = number of window starts j < i such that row[j : j+width] is all free (cum has len(row)+1 entries)."""
    S = len(row)
    cum = np.zeros(S + 1, np.int64)
    if 0 < width <= S:
        run = np.concatenate(([0], np.cumsum(row != 0)))
        ok = (run[width:] - run[:-width]) == width            # window j .. j+width-1 free, j = 0 .. S-width
        cum[1:S - width + 2] = np.cumsum(ok)
        cum[S - width + 2:] = cum[S - width + 1]
    return cum


def _calculate_allocation_possibilities(available_slots: np.ndarray, required_slots: int) -> int:
    """Number of placements of `required_slots` contiguous slots inside the free runs of a row (reference :629-645)."""
    if required_slots <= 0:
        return 0
    return int(_window_counts(np.asarray(available_slots), int(required_slots))[-1])


def heuristic_mscl_plugin(env):
    """Minimum spectrum capacity loss, reference :647-749, written against the plugin API: among all (route, modulation, start) whose GSNR passes, the
    one that destroys the fewest placements — summed over the configured bit rates (at the candidate's modulation) and
    over every route of the network sharing a link with the candidate route — when its slots (no guard) are taken.
    The loss of blocking [a, b) on a row = the free windows of that width starting in (a - width, b): a difference of
    prefix counts, so the whole search is a few vector operations on the grid view."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    S = sim_env.num_spectrum_resources
    grid = np.asarray(sim_env.topology.graph["available_slots"])
    every_route = [p for pair in sim_env.k_shortest_paths for p in sim_env.k_shortest_paths[pair]]   # both directions
    link_ids = lambda path: [sim_env.topology[l.node1][l.node2]["index"] for l in path.links]     # noqa: E731
    route_rows = {}
    options, cands = [], []
    no_slots = low_osnr = False
    for path_idx, path in enumerate(_routes(sim_env)):
        mine = set(link_ids(path))
        touching = [p for p in every_route if mine.intersection(link_ids(p))]
        avail = sim_env.get_available_slots(path)
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            if slots <= 0:
                continue
            starts = sim_env._get_candidates(avail, slots, S)
            if not starts:
                no_slots = True
                continue
            widths = [sim_env.get_number_slots(_RateOnly(rate), modulation) for rate in sim_env.bit_rates]
            totals = []
            for width in widths:                      # sum over touching routes of the prefix window counts
                acc = np.zeros(S + 1, np.int64)
                if width > 0:
                    for p in touching:
                        key = (p.id, width)
                        if key not in route_rows:
                            route_rows[key] = _window_counts(np.prod(grid[link_ids(p), :], axis=0), width)
                        acc += route_rows[key]
                totals.append(acc)
            for start in starts:
                loss = 0
                for width, acc in zip(widths, totals):
                    if width > 0:
                        loss += int(acc[min(start + slots, S)] - acc[max(0, start - width + 1)])
                options.append((loss, path_idx, modulation_idx, modulation, start))
                cands.append((path, start, slots))
    best, best_loss = None, math.inf
    if cands:
        osnr = sim_env.calculate_osnr_many(cands)[:, 0]
        for (loss, path_idx, modulation_idx, modulation, start), value in zip(options, osnr):
            if not _passes(sim_env, modulation, value):
                low_osnr = True
            elif loss < best_loss:
                best, best_loss = get_action_index(sim_env, path_idx, modulation_idx, start), loss
    if best is not None:
        return best, False, False
    return sim_env.action_space.n - 1, no_slots, low_osnr


class _RateOnly:
    """What `get_number_slots` reads of a service."""
    def __init__(self, bit_rate):
        self.bit_rate = bit_rate


def _get_largest_contiguous_block(available_slots: np.ndarray) -> int:
    """Longest free run of a row (reference :751-763)."""
    row = np.asarray(available_slots)
    if not np.any(row):
        return 0
    _, values, lengths = rle(row)
    free = lengths[values == 1]
    return int(free.max()) if free.size else 0


def _simplified_mscl_scores(sim_env, routes=None):
    """(score, action) of the first-fit candidate of every (route, modulation) that passes QoT; score = the longest free
    run left on the route once the candidate's slots (no guard) are taken.  Also yields the blocking flags."""
    for path_idx, path, modulation_idx, modulation, slots, avail, starts in _first_fit_options(sim_env, routes):
        if not starts:
            yield path_idx, None, None, True, False
            continue
        start = starts[0]
        if _passes(sim_env, modulation, _osnr_at(sim_env, path, modulation, start, slots)):
            left = np.array(avail, copy=True)
            left[start:start + slots] = 0
            yield path_idx, _get_largest_contiguous_block(left), get_action_index(sim_env, path_idx, modulation_idx, start), False, False
        else:
            yield path_idx, None, None, False, True


def heuristic_mscl_simplified_plugin(env):
    """Reference :765-839 — over all routes and modulations, the first-fit candidate leaving the longest free run."""
    sim_env = get_qrmsa_env(env)
    best, best_score = None, -1
    no_slots = low_osnr = False
    for _, score, action, a, b in _simplified_mscl_scores(sim_env):
        no_slots, low_osnr = no_slots or a, low_osnr or b
        if action is not None and score > best_score:
            best, best_score = action, score
    if best is not None:
        return best, False, False
    return sim_env.action_space.n - 1, no_slots, low_osnr


def heuristic_mscl_sequential_simplified_plugin(env):
    """Reference :841-921 — the same score, but the first route with any passing candidate answers."""
    sim_env = get_qrmsa_env(env)
    no_slots = low_osnr = False
    for path_idx, path in enumerate(_routes(sim_env)):
        best, best_score = None, -1
        for _, score, action, a, b in _simplified_mscl_scores(sim_env, [(path_idx, path)]):
            no_slots, low_osnr = no_slots or a, low_osnr or b
            if action is not None and score > best_score:
                best, best_score = action, score
        if best is not None:
            return best, False, False
    return sim_env.action_space.n - 1, no_slots, low_osnr


def heuristic_psr_plugin(env, variant: str = "O", coef_dist: float = 1.0, coef_slots: float = 1.0):
    """Power-series routing, reference :1019-1119.  The route cost (sum for 'C', product otherwise, over the route's
    links of coef_dist * length/longest and coef_slots / (placements of the most efficient format + 1)) only gates a
    route against `best_cost`, which stays infinite until the search ends at the first success — so the answer is the
    first route, best modulation, lowest start (ALL starts of a modulation are tried) whose GSNR passes."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    best_cost = float("inf")
    for path_idx, path in enumerate(_routes(sim_env)):
        cost = 0.0 if variant == "C" else 1.0
        longest = max(link.length for link in path.links) if path.links else 1.0
        avail = sim_env.get_available_slots(path)
        width = sim_env.get_number_slots(service, sim_env.modulations[-1])
        x2 = 1.0 / (_calculate_allocation_possibilities(avail, width) + 1)
        for link in path.links:
            x1 = link.length / longest if longest > 0 else 1.0
            if variant == "C":
                cost += coef_dist * x1 + coef_slots * x2
            else:
                cost *= (coef_dist * x1) * (coef_slots * x2)
        if not cost < best_cost:
            continue
        for _, _, modulation_idx, modulation, slots, _, starts in _first_fit_options(sim_env, [(path_idx, path)]):
            if not starts:
                continue
            osnr = sim_env.calculate_osnr_many([(path, start, slots) for start in starts])[:, 0]
            for start, value in zip(starts, osnr):
                if _passes(sim_env, modulation, value):
                    return get_action_index(sim_env, path_idx, modulation_idx, start), False, False
    return sim_env.action_space.n - 1, True, False


def heuristic_exact_fit_plugin(env):
    """Reference :1121-1227 — per (route, modulation): the first free run of EXACTLY the needed length, else the
    smallest run that is long enough (first among equals); then the QoT check.  No guard slot is asked for, so the env may
    answer such an action with the occupied-slots penalty (qrmsa.pyx:886-897)."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    no_slots = low_osnr = False
    for path_idx, path in enumerate(_routes(sim_env)):
        avail = sim_env.get_available_slots(path)
        starts, values, lengths = rle(avail)
        free = np.flatnonzero(values == 1)
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            if slots <= 0:
                continue
            if free.size == 0:
                no_slots = True
                continue
            exact = free[lengths[free] == slots]
            if exact.size:
                slot = int(starts[exact[0]])
            else:
                enough = free[lengths[free] >= slots]
                if enough.size == 0:
                    no_slots = True
                    continue
                slot = int(starts[enough[np.argmin(lengths[enough])]])      # argmin: first of the smallest
            if _passes(sim_env, modulation, _osnr_at(sim_env, path, modulation, slot, slots)):
                return get_action_index(sim_env, path_idx, modulation_idx, slot), False, False
            low_osnr = True
    return env.action_space.n - 1, (no_slots and not low_osnr), low_osnr


def heuristic_from_mask(env, mask: np.ndarray) -> int:
    """Reference :76-198 is a debugging aid: it re-derives every entry of an action mask through the plugin API
    (candidates + GSNR), reports disagreements, and returns a uniformly random action index.  Same here, without the
    per-action printing; disagreements raise AssertionError."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    mask = np.asarray(mask)
    S = sim_env.num_spectrum_resources
    assert mask[sim_env.action_space.n - 1] == 1, "the reject action must always be allowed"
    want = np.zeros(len(mask), mask.dtype)
    want[-1] = 1
    entries, cands = [], []
    for path_idx, path in enumerate(_routes(sim_env)):
        avail = sim_env.get_available_slots(path)
        for column in range(sim_env.modulations_to_consider):
            modulation_idx = sim_env.encoded_decimal_to_array((path_idx * sim_env.modulations_to_consider + column) * S)[1]
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            for start in (sim_env._get_candidates(avail, slots, S) if slots > 0 else []):
                entries.append(((path_idx * sim_env.modulations_to_consider + column) * S + start, modulation))
                cands.append((path, start, slots))
    if cands:
        osnr = sim_env.calculate_osnr_many(cands)[:, 0]
        for (index, modulation), value in zip(entries, osnr):
            want[index] = 1 if _passes(sim_env, modulation, value) else 0
    wrong = np.flatnonzero(want != mask)
    assert wrong.size == 0, f"mask disagrees with the plugin-API derivation at actions {wrong[:8].tolist()}"
    return int(np.random.choice(len(mask)))


def heuristic_highest_snr_plugin(env):
    """The same policy written against the plugin API only (slow: one device query per candidate)."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    best, best_osnr = None, float("-inf")
    no_slots = low_osnr = False
    for path_idx, path in enumerate(sim_env.k_shortest_paths[service.source, service.destination]):
        avail = sim_env.get_available_slots(path)
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            if slots <= 0:
                continue
            starts = sim_env._get_candidates(avail, slots, sim_env.num_spectrum_resources)
            if not starts:
                no_slots = True
                continue
            for start in starts:
                _stage_candidate(sim_env, service, path, modulation, start, slots)
                osnr, _, _ = calculate_osnr(sim_env, service)
                if osnr >= modulation.minimum_osnr + sim_env.margin:
                    if osnr > best_osnr:
                        best, best_osnr = get_action_index(sim_env, path_idx, modulation_idx, start), osnr
                else:
                    low_osnr = True
    if best is None:
        return env.action_space.n - 1, (no_slots and not low_osnr), low_osnr
    return best, False, False


# ----------------------------------------------------------------------------------------------------------------------
# Public names -> the policies fused on device (one launch per decision; the same ids drive whole batched episodes through
# BatchedQRMSAEnv.step_policy(policy=...)).  The `*_plugin` bodies above are the same policies written against the plugin
# API only; tests hold both to the decisions captured from the reference.
# ----------------------------------------------------------------------------------------------------------------------
def _fused(policy_id, doc):
    def call(env):
        return get_qrmsa_env(env).policy_action(policy_id)
    call.__doc__ = doc
    return call


shortest_available_path_lowest_spectrum_best_modulation = _fused(
    _nat.POLICY_LOWEST_SPECTRUM, "Reference :431-490, fused on device (ONGYM_POLICY_LOWEST_SPECTRUM).")
best_modulation_load_balancing = _fused(_nat.POLICY_BEST_MOD_LB, "Reference :491-545, fused on device.")
heuristic_load_balancing_first_fit = _fused(_nat.POLICY_LB_FIRST_FIT, "Reference :202-269, fused on device.")
heuristic_mscl_simplified = _fused(_nat.POLICY_MSCL_SIMPLIFIED, "Reference :765-839, fused on device.")
heuristic_mscl_sequential_simplified = _fused(_nat.POLICY_MSCL_SEQUENTIAL, "Reference :841-921, fused on device.")
heuristic_exact_fit = _fused(_nat.POLICY_EXACT_FIT, "Reference :1121-1227, fused on device.")
heuristic_lowest_fragmentation = _fused(
    _nat.POLICY_LOWEST_FRAGMENTATION,
    "Reference :330-414, fused on device (csrc/ongym_scored.hpp): the float score is reproduced bit for bit from a table of "
    "p*log(p) and the reference's own summation order.")


def heuristic_mscl(env):
    """Reference :647-749, fused on device (csrc/ongym_scored.hpp) for discrete bit rates — the capacity loss is summed over
    `env.bit_rates`; with continuous bit-rate selection the plugin body answers."""
    sim_env = get_qrmsa_env(env)
    if sim_env.bit_rate_selection != "discrete":
        return heuristic_mscl_plugin(env)
    return sim_env.policy_action(_nat.POLICY_MSCL)


def heuristic_psr(env, variant: str = "O", coef_dist: float = 1.0, coef_slots: float = 1.0):
    """Reference :1019-1119.  With finite positive coefficients the route cost never changes the answer (see
    `heuristic_psr_plugin`), which is then the fused device policy; anything else takes the literal plugin path."""
    import math as _m
    if all(_m.isfinite(v) and v > 0 for v in (coef_dist, coef_slots)):
        return get_qrmsa_env(env).policy_action(_nat.POLICY_PSR)
    return heuristic_psr_plugin(env, variant, coef_dist, coef_slots)
