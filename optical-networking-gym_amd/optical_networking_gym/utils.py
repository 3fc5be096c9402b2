"""Small array helpers kept for plugin compatibility (reference: optical_networking_gym/utils.pyx:44-58).

`link_shannon_entropy_`, `fragmentation_route_cuts`, `fragmentation_route_rss` (utils.pyx:61-110) feed the
lowest-fragmentation plugin; like the reference they measure the runs of ZEROS of the rows they are given.

`rle` is NOT on the device hot path — the kernels scan the packed free-slot bitmap directly
(csrc/ongym_device.hpp: run_and / first_set); it exists because plugin heuristics call it on rows they obtained from
`env.get_available_slots`.
"""
from __future__ import annotations

import math

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


def _zero_runs(row) -> np.ndarray:
    """Lengths of the runs of zeros of `row`, left to right."""
    _, values, lengths = rle(row)
    if values is None:
        return np.zeros(0, np.int64)
    return lengths[values == 0]


def link_shannon_entropy_(link) -> float:
    """-sum p*ln(p) over the zero-runs of one link row, p = run length / row length (accumulated left to right)."""
    runs = _zero_runs(np.array(link))
    total = len(link)
    acc = 0.0
    for run in runs:
        if run == 0:
            continue
        p = run / total
        acc += p * math.log(p)
    return -acc if acc != 0 else 0.0


def fragmentation_route_cuts(path_spectrums) -> int:
    """Number of zero-runs summed over the rows of a route."""
    return int(sum(len(_zero_runs(row)) for row in path_spectrums))


def fragmentation_route_rss(path_spectrums) -> float:
    """sqrt(sum of squared zero-run lengths) / (sum of zero-run lengths) over the rows of a route; 0 if there is none."""
    squares = 0.0
    length = 0.0
    for row in path_spectrums:
        runs = _zero_runs(row)
        if runs.size == 0:
            continue
        squares += np.sum(runs.astype(float) ** 2)
        length += np.sum(runs)
    if length == 0:
        return 0.0
    return math.sqrt(squares) / length
