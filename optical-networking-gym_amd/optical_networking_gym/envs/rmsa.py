"""`RMSAEnv` placeholder with the reference's constructor surface (optical_networking_gym/envs/rmsa.pyx:44-81).

In the reference this class is a stub — it allocates two bookkeeping arrays and its step/reset are placeholders; it is
kept only so that BASELINE config 1 ("RMSA env ... plumbing, no GPU", tests/test_rmsa.py:38-96 of the reference)
constructs. There is no compute here and none on the device.
"""
from __future__ import annotations

import numpy as np


class RMSAEnv:
    def __init__(self, topology, num_spectrum_resources: int):
        self.topology = topology
        self.num_spectrum_resources = int(num_spectrum_resources)
        self.bit_rates = (0, 40, 100)
        edges = topology.number_of_edges()
        self.spectrum_use = np.full((edges, self.num_spectrum_resources), -1, dtype=np.int32)
        self.spectrum_allocation = np.full((edges, self.num_spectrum_resources), -1, dtype=np.int64)

    def reset(self, *, seed=None, options=None):
        return None

    def step(self, action):
        raise NotImplementedError("RMSAEnv is a placeholder in the reference as well; use QRMSAEnv")

    def close(self):
        return None
