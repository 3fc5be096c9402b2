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

from typing import Optional

import numpy as np

from .. import _native as _nat
from ..core.osnr import calculate_osnr
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


def _not_built(name, where):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"{name} (reference heuristics.py:{where}) is not built yet; write it against the plugin "
                                  f"API (get_available_slots / _get_candidates / calculate_osnr) or use a fused policy")
    stub.__name__ = name
    return stub


# names the reference's example scripts import (graph_load.py:80-90, graph_launch_power.py:64-78); importing them works,
# calling one that is not built fails loudly
heuristic_from_mask = _not_built("heuristic_from_mask", "76-198")
heuristic_load_balancing_first_fit = _not_built("heuristic_load_balancing_first_fit", "202-269")
heuristic_lowest_fragmentation = _not_built("heuristic_lowest_fragmentation", "330-414")
shortest_available_path_lowest_spectrum_best_modulation = _not_built(
    "shortest_available_path_lowest_spectrum_best_modulation", "431-490")
best_modulation_load_balancing = _not_built("best_modulation_load_balancing", "491-545")
heuristic_mscl = _not_built("heuristic_mscl", "647-749")
heuristic_mscl_simplified = _not_built("heuristic_mscl_simplified", "765-839")
heuristic_mscl_sequential_simplified = _not_built("heuristic_mscl_sequential_simplified", "841-921")
heuristic_psr = _not_built("heuristic_psr", "1019-1119")
heuristic_exact_fit = _not_built("heuristic_exact_fit", "1121-1227")


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
