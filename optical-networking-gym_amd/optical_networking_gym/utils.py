"""Small array helpers kept for plugin compatibility (reference: optical_networking_gym/utils.pyx:44-58).

`rle` is NOT on the device hot path — the kernels scan the packed free-slot bitmap directly
(csrc/ongym_device.hpp: run_and / first_set); it exists because plugin heuristics call it on rows they obtained from
`env.get_available_slots`.
"""
from __future__ import annotations

import numpy as np


def rle(inarray):
    """Run-length encoding: returns (start positions, run values, run lengths); (None, None, None) for empty input."""
    values = np.asarray(inarray)
    size = values.shape[0] if values.ndim else 0
    if size == 0:
        return (None, None, None)
    boundaries = np.flatnonzero(values[1:] != values[:-1]) + 1          # first index of every run but the first
    starts = np.concatenate(([0], boundaries))
    lengths = np.diff(np.concatenate((starts, [size])))
    return starts, values[starts], lengths
