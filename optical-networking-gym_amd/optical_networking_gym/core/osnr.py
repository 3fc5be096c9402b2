"""GN-model QoT estimate, plugin entry point (reference: optical_networking_gym/core/osnr.pyx:21-142).

The arithmetic runs on the GPU (csrc/ongym_device.hpp: gn_build_list / gn_eval); this function only forwards the
candidate lightpath `(service.path, service.initial_slot, service.number_slots)` of the plugin's mutated
`current_service` to `ongym_query_gsnr`.  Centre frequency and bandwidth are re-derived from slot and slot count on
device exactly as the plugins compute them (heuristics.py:947-952), launch power is the env's.
"""
from __future__ import annotations


def calculate_osnr(env, current_service):
    """Returns (gsnr_dB, ase_dB, nli_dB) of `current_service` against the running services of `env`."""
    inner = env
    while not hasattr(inner, "calculate_osnr"):
        inner = inner.env
    return inner.calculate_osnr(current_service)
