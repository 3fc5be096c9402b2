"""Twin of the reference's traffic draws on CPython's `random` (qrmsa.pyx:1079-1089, 1134-1148): used to pin the draw
order and float32 rounding points against the captured request streams."""
import random

import numpy as np

from optical_networking_gym._native import REQUEST_DTYPE


def cpython_request_stream(seed, n, n_nodes, load, mean_holding, bit_rates=None, lo=25, hi=100):
    rng = random.Random(seed)
    nodes = list(range(n_nodes))
    w = np.full(n_nodes, 1.0 / n_nodes)
    mean_iat = 1 / (load / mean_holding)
    out = np.zeros(n, REQUEST_DTYPE)
    current_time = 0.0
    for i in range(n):
        at = np.float32(current_time + rng.expovariate(1 / mean_iat))
        current_time = float(at)
        ht = np.float32(rng.expovariate(1.0 / mean_holding))
        src = rng.choices(nodes, weights=w)[0]
        w2 = np.copy(w)
        w2[src] = 0.0
        w2 /= np.sum(w2)
        dst = rng.choices(nodes, weights=w2)[0]
        if bit_rates is not None:
            br = rng.choices(bit_rates, [1.0 / len(bit_rates)] * len(bit_rates), k=1)[0]
        else:
            br = rng.randint(lo, hi)
        out[i] = (at, ht, np.float32(br), src, dst)
    return out
