"""Policy plugins with the reference's calling convention: `f(env) -> (action, blocked_resources, blocked_osnr)`.

Reference: optical_networking_gym/heuristics/heuristics.py — `get_qrmsa_env` (:15-33), `get_action_index` (:36-54),
`heuristic_shortest_available_path_first_fit_best_modulation` (:923-966), `heuristic_highest_snr` (:272-328).

* The benchmark default (first fit) is answered by the policy fused on device (`ongym_policy_actions`).
* `heuristic_shortest_available_path_first_fit_best_modulation_plugin` is the same policy written against the plugin API
  only (k_shortest_paths / get_number_slots / get_available_slots / _get_candidates / calculate_osnr): it exists to show
  — and test — that plugins written for the reference run unchanged on the compatibility view.
"""
from __future__ import annotations

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
    """Among every (path, modulation) pair's first-fit candidate that clears its threshold, take the highest GSNR."""
    sim_env = get_qrmsa_env(env)
    service = sim_env.current_service
    best, best_osnr = None, float("-inf")
    no_slots = low_osnr = False
    for path_idx, path in enumerate(sim_env.k_shortest_paths[service.source, service.destination]):
        avail = sim_env.get_available_slots(path)
        for modulation_idx in range(sim_env.max_modulation_idx, -1, -1):
            modulation = sim_env.modulations[modulation_idx]
            slots = sim_env.get_number_slots(service, modulation)
            starts = sim_env._get_candidates(avail, slots, sim_env.num_spectrum_resources) if slots > 0 else []
            if not starts:
                no_slots = True
                continue
            _stage_candidate(sim_env, service, path, modulation, starts[0], slots)
            osnr, _, _ = calculate_osnr(sim_env, service)
            if osnr >= modulation.minimum_osnr + sim_env.margin:
                if osnr > best_osnr:
                    best, best_osnr = get_action_index(sim_env, path_idx, modulation_idx, starts[0]), osnr
            else:
                low_osnr = True
    if best is None:
        return env.action_space.n - 1, no_slots, low_osnr
    return best, False, False
