"""Shared helpers for the test-suite: golden fixture loading and config construction."""
from __future__ import annotations

import json
import os

import numpy as np

from optical_networking_gym._native import ConfigHolder, REQUEST_DTYPE
from optical_networking_gym._tables import StaticTables
from optical_networking_gym.topology import Modulation

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TABLE_FILE = {"nsfnet": "tables_nsfnet.json", "nobel-eu": "tables_nobel-eu.json", "cost239": "tables_cost239.json",
              "ring4": "tables_ring4.json", "germany50": "tables_germany50.json"}


def jocn_modulations():
    """Thresholds of examples/JOCN_Benchmark_2024/graph_load.py:252-295 (reference)."""
    return (Modulation("BPSK", 100000, 1, 3.71, -14), Modulation("QPSK", 2000, 2, 6.72, -17),
            Modulation("8QAM", 1000, 3, 10.84, -20), Modulation("16QAM", 500, 4, 13.24, -23),
            Modulation("32QAM", 250, 5, 16.16, -26), Modulation("64QAM", 125, 6, 19.01, -29))


_tables_cache = {}


def golden_tables(name: str) -> StaticTables:
    if name not in _tables_cache:
        with open(os.path.join(GOLDEN, TABLE_FILE[name])) as f:
            _tables_cache[name] = StaticTables.from_golden(json.load(f))
    return _tables_cache[name]


def load_traj(tag: str):
    meta = json.load(open(os.path.join(GOLDEN, tag + ".json")))
    data = np.load(os.path.join(GOLDEN, tag + ".npz"))
    return meta, data


def traj_requests(data) -> np.ndarray:
    n = len(data["req_at"])
    reqs = np.zeros(n, REQUEST_DTYPE)
    reqs["arrival_time"] = data["req_at"]
    reqs["holding_time"] = data["req_ht"]
    reqs["bit_rate"] = data["req_br"]
    reqs["source"] = data["req_src"]
    reqs["destination"] = data["req_dst"]
    return reqs


def holder_for(meta: dict, batch: int = 1, capacity: int = 1024, **over) -> ConfigHolder:
    kw = dict(modulations=jocn_modulations(), num_spectrum_resources=meta["S"], batch=batch, capacity=capacity,
              episode_length=meta["episode_length"], load=meta["load"], mean_service_holding_time=meta["mean_holding"],
              bit_rate_selection=meta["bit_rate_selection"], bit_rates=tuple(meta["bit_rates"]),
              bit_rate_lower_bound=25, bit_rate_higher_bound=100, launch_power_dbm=meta["launch_power_dbm"],
              frequency_start=meta["frequency_start"], frequency_slot_bandwidth=meta["slot_bw"],
              margin=meta["margin"], nslots_channel_width=meta.get("nslots_channel_width", 0.0))
    kw.update(over)
    tables = golden_tables(meta["topology"])
    if meta.get("k_paths", tables.k_paths) < tables.k_paths:     # fixtures captured with fewer candidate routes per pair
        tables = tables.truncated(meta["k_paths"])
    return ConfigHolder(tables, **kw)


ALL_TRAJ = ["traj_nsfnet320", "traj_nsfnet320_hi", "traj_nobeleu320", "traj_nsfnet768", "traj_cost239",
            "traj_nsfnet320_cont", "traj_ring4"]
