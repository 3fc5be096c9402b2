"""ctypes mirror of include/ongym.h and the loader of libongym_hip.so.

There is no CPU fallback: if the HIP library is missing or fails to load, `load_library()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from ._tables import StaticTables, cumulative, modulation_arrays

ABI_VERSION = 3
PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(os.path.dirname(PKG_DIR), "csrc")
HIP_LIB_PATH = os.path.join(CSRC_DIR, "libongym_hip.so")

POLICY_FIRST_FIT = 0
POLICY_LOAD_BALANCING = 1
POLICY_HIGHEST_SNR = 2
POLICY_LOWEST_SPECTRUM, POLICY_LB_FIRST_FIT, POLICY_BEST_MOD_LB = 3, 4, 5
POLICY_MSCL_SIMPLIFIED, POLICY_MSCL_SEQUENTIAL, POLICY_PSR, POLICY_EXACT_FIT = 6, 7, 8, 9
POLICY_LOWEST_FRAGMENTATION, POLICY_MSCL = 10, 11
F_BLOCKED_RESOURCES, F_BLOCKED_OSNR, F_QOT_ERROR, F_OVERFLOW, F_NO_REQUEST = 1, 2, 4, 8, 16

_i32p, _f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)


class OngymConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("abi_version", C.c_int32),
        ("n_nodes", C.c_int32), ("n_links", C.c_int32), ("n_paths", C.c_int32), ("k_paths", C.c_int32),
        ("max_hops", C.c_int32), ("n_mods", C.c_int32), ("n_slots", C.c_int32),
        ("batch", C.c_int32), ("capacity", C.c_int32), ("episode_length", C.c_int32), ("auto_reset", C.c_int32),
        ("bit_rate_mode", C.c_int32), ("n_bit_rates", C.c_int32), ("bit_rate_lo", C.c_int32),
        ("bit_rate_hi", C.c_int32), ("device", C.c_int32), ("io_device", C.c_int32), ("measure_disruptions", C.c_int32),
        ("defragmentation", C.c_int32), ("n_defrag_services", C.c_int32),
        ("frequency_start", C.c_double), ("slot_bandwidth", C.c_double), ("channel_width", C.c_double),
        ("launch_power_w", C.c_double), ("margin", C.c_double), ("load", C.c_double),
        ("mean_holding_time", C.c_double),
        ("pair_paths", _i32p), ("path_hops", _i32p), ("path_links", _i32p), ("link_nspans", _i32p),
        ("link_span_km", _f64p), ("link_alpha", _f64p), ("link_nf", _f64p),
        ("mod_se", _i32p), ("mod_min_osnr", _f64p), ("bit_rates", _f64p), ("bit_rate_cum", _f64p),
        ("node_cum", _f64p),
        ("replica_launch_power_w", _f64p), ("replica_load", _f64p), ("replica_margin", _f64p),
        ("path_len_norm", _f64p), ("max_bit_rate", C.c_double),
        ("track_service_ids", C.c_int32), ("n_mods_consider", C.c_int32),
        ("nslots_channel_width", C.c_double),
    ]


REQUEST_DTYPE = np.dtype([("arrival_time", "<f4"), ("holding_time", "<f4"), ("bit_rate", "<f4"),
                          ("source", "<i2"), ("destination", "<i2")])
STEP_DTYPE = np.dtype([("action", "<i4"), ("route", "<i2"), ("modulation", "<i2"), ("slot", "<i2"),
                       ("nslots", "<i2"), ("accepted", "u1"), ("terminated", "u1"), ("retry", "u1"),
                       ("flags", "u1"), ("active", "<i4"), ("osnr", "<f8"), ("ase", "<f8"), ("nli", "<f8"),
                       ("reward", "<f8")], align=True)
SERVICE_DTYPE = np.dtype([("path_id", "<i4"), ("slot", "<i2"), ("nslots", "<i2"), ("modulation", "<i2"),
                          ("reserved", "<i2"), ("release_time", "<f4"), ("service_id", "<i4"), ("pad_", "<i4"),
                          ("osnr", "<f8")], align=True)
MOVE_LOG = 64
MOVE_DTYPE = np.dtype([("service_id", "<i4"), ("slot", "<i4"), ("osnr", "<f8"), ("ase", "<f8"), ("nli", "<f8")], align=True)
STATS_DTYPE = np.dtype([
    ("services_processed", "<i8"), ("services_accepted", "<i8"),
    ("episode_services_processed", "<i8"), ("episode_services_accepted", "<i8"),
    ("bit_rate_requested", "<f8"), ("bit_rate_provisioned", "<f8"),
    ("episode_bit_rate_requested", "<f8"), ("episode_bit_rate_provisioned", "<f8"),
    ("rejected", "<i8"), ("episode_modulation_hist", "<i8", (8,)), ("episode_osnr_sum", "<f8"),
    ("episodes_completed", "<i8"), ("disrupted_services", "<i8"), ("episode_disrupted_services", "<i8"),
    ("episode_defrag_cycles", "<i8"), ("episode_service_reallocations", "<i8"),
    ("step_defrag_cycles", "<i8"), ("step_service_reallocations", "<i8"),
    ("total_steps", "<i8"), ("total_accepted", "<i8"), ("total_gn_evals", "<i8"),
    ("total_interferer_terms", "<i8"), ("total_paths_tried", "<i8"), ("total_path_hops", "<i8"),
    ("total_gn_shortcuts", "<i8"), ("total_active_sum", "<i8"), ("current_time", "<f8"), ("active", "<i4"), ("flags", "<i4"),
    ("max_modulation_idx", "<i4"), ("reserved0_", "<i4"),
    # terminal-step snapshot (kept last, see include/ongym.h)
    ("last_episode_processed", "<i8"), ("last_episode_accepted", "<i8"), ("last_rejected", "<i8"),
    ("last_service_blocking_rate", "<f8"), ("last_episode_service_blocking_rate", "<f8"),
    ("last_bit_rate_blocking_rate", "<f8"), ("last_episode_bit_rate_blocking_rate", "<f8"),
    ("last_modulation_hist", "<i8", (8,)), ("last_mean_gsnr", "<f8"), ("last_episode_disrupted", "<i8"),
    ("last_episode_defrag_cycles", "<i8"), ("last_episode_service_reallocations", "<i8")], align=True)


class ConfigHolder:
    """An `OngymConfig` plus the numpy arrays its pointers refer to (kept alive here)."""

    def __init__(self, tables: StaticTables, *, modulations: Sequence, num_spectrum_resources: int = 320,
                 batch: int = 1, capacity: int = 1024, episode_length: int = 1000, auto_reset: bool = False,
                 load: float = 10.0, mean_service_holding_time: float = 10800.0,
                 bit_rate_selection: str = "continuous", bit_rates: Sequence[float] = (10, 40, 100),
                 bit_rate_probabilities: Optional[Sequence[float]] = None,
                 node_request_probabilities: Optional[Sequence[float]] = None,
                 bit_rate_lower_bound: float = 25.0, bit_rate_higher_bound: float = 100.0,
                 launch_power_dbm: float = 0.0, frequency_start: float = 3e8 / 1565e-9,
                 frequency_slot_bandwidth: float = 12.5e9, margin: float = 0.0, channel_width: float = 12.5,
                 nslots_channel_width: float = 0.0,
                 device: int = 0, io_device: bool = False, measure_disruptions: bool = False,
                 defragmentation: bool = False, n_defrag_services: int = 0,
                 replica_launch_power_dbm: Optional[Sequence[float]] = None,
                 replica_load: Optional[Sequence[float]] = None,
                 replica_margin: Optional[Sequence[float]] = None, track_service_ids: bool = False,
                 modulations_to_consider: Optional[int] = None):
        if capacity % 64 or capacity <= 0:
            raise ValueError("capacity must be a positive multiple of 64")
        if load <= 0 or mean_service_holding_time <= 0:
            raise ValueError("Both load and mean_service_holding_time must be positive values.")
        self.tables = tables
        t = tables
        mod_se, mod_thr = modulation_arrays(modulations)
        keep = self._keep = {}

        def i32(name, a):
            keep[name] = np.ascontiguousarray(a, np.int32)
            return keep[name].ctypes.data_as(_i32p)

        def f64(name, a):
            keep[name] = np.ascontiguousarray(a, np.float64)
            return keep[name].ctypes.data_as(_f64p)

        c = self.struct = OngymConfig()
        c.struct_size = C.sizeof(OngymConfig)
        c.abi_version = ABI_VERSION
        c.n_nodes, c.n_links, c.n_paths, c.k_paths = t.n_nodes, t.n_links, t.n_paths, t.k_paths
        c.max_hops, c.n_mods, c.n_slots = t.max_hops, len(mod_se), int(num_spectrum_resources)
        c.batch, c.capacity, c.episode_length, c.auto_reset = int(batch), int(capacity), int(episode_length), int(auto_reset)
        if bit_rate_selection == "discrete":
            c.bit_rate_mode = 0
            rates = np.asarray(bit_rates, np.float64)
            cum = cumulative(bit_rate_probabilities, len(rates))
        elif bit_rate_selection == "continuous":
            c.bit_rate_mode = 1
            rates = np.asarray([0.0])
            cum = np.asarray([1.0])
        else:
            raise ValueError("bit_rate_selection must be 'continuous' or 'discrete'")
        c.n_bit_rates = len(rates)
        c.bit_rate_lo, c.bit_rate_hi = int(bit_rate_lower_bound), int(bit_rate_higher_bound)  # qrmsa.pyx:250-254
        c.device, c.io_device = int(device), int(bool(io_device))
        c.measure_disruptions = int(bool(measure_disruptions))
        c.defragmentation, c.n_defrag_services = int(bool(defragmentation)), int(n_defrag_services)
        c.track_service_ids = int(bool(track_service_ids))
        mtc = len(mod_se) if modulations_to_consider is None else min(int(modulations_to_consider), len(mod_se))   # qrmsa.pyx:313
        if mtc < 1:
            raise ValueError("modulations_to_consider must be >= 1")
        c.n_mods_consider = mtc
        c.frequency_start, c.slot_bandwidth = float(frequency_start), float(frequency_slot_bandwidth)
        c.channel_width = float(channel_width)
        c.nslots_channel_width = float(nslots_channel_width)      # `bands`: the C band's width in Hz (quirk Q9); 0 = channel_width
        c.launch_power_w = 10 ** ((float(launch_power_dbm) - 30) / 10)  # qrmsa.pyx:288
        c.margin, c.load, c.mean_holding_time = float(margin), float(load), float(mean_service_holding_time)
        c.pair_paths = i32("pair_paths", t.pair_paths)
        c.path_hops = i32("path_hops", t.path_hops)
        c.path_links = i32("path_links", t.path_links)
        c.link_nspans = i32("link_nspans", t.link_nspans)
        c.link_span_km = f64("link_span_km", t.link_span_km)
        c.link_alpha = f64("link_alpha", t.link_alpha)
        c.link_nf = f64("link_nf", t.link_nf)
        c.mod_se = i32("mod_se", mod_se)
        c.mod_min_osnr = f64("mod_min_osnr", mod_thr)
        c.bit_rates = f64("bit_rates", rates)
        c.bit_rate_cum = f64("bit_rate_cum", cum)
        c.node_cum = f64("node_cum", cumulative(node_request_probabilities, t.n_nodes))
        if replica_launch_power_dbm is not None:
            c.replica_launch_power_w = f64("rlp", [10 ** ((float(x) - 30) / 10) for x in replica_launch_power_dbm])
        if replica_load is not None:
            c.replica_load = f64("rload", replica_load)
        if replica_margin is not None:
            c.replica_margin = f64("rmargin", replica_margin)
        # observation(): route lengths normalised by min/max LINK length (qrmsa.pyx:692-705), max(bit_rates) (:679)
        lo, hi = float(np.min(t.link_length)), float(np.max(t.link_length))
        c.path_len_norm = f64("path_len_norm", [(x - lo) / (hi - lo) if hi != lo else 0.0 for x in t.path_length])
        # observation() normalises by max(self.bit_rates) whatever the selection mode (qrmsa.pyx:679, 688): in "continuous"
        # mode that is the max of the (otherwise unused) bit_rates tuple
        c.max_bit_rate = float(np.max(np.asarray(bit_rates, np.float64))) if len(bit_rates) else 0.0
        for name in ("rlp", "rload", "rmargin"):
            if name in keep and len(keep[name]) != batch:
                raise ValueError("per-replica override arrays must have `batch` entries")
        self.mod_se, self.mod_thr = mod_se, mod_thr
        self.bit_rates = rates

    @property
    def reject_action(self) -> int:
        c = self.struct
        return c.k_paths * c.n_mods_consider * c.n_slots


_LIB = None


def _declare(lib):
    vp = C.c_void_p
    lib.ongym_create.argtypes = [C.POINTER(OngymConfig), C.POINTER(vp)]
    lib.ongym_destroy.argtypes = [vp]
    lib.ongym_destroy.restype = None
    lib.ongym_seed.argtypes = [vp, C.c_uint64]
    lib.ongym_seed_base.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.ongym_set_requests.argtypes = [vp, vp, C.c_int64]
    lib.ongym_reset.argtypes = [vp, vp]
    lib.ongym_reset_episode_counters.argtypes = [vp, vp]
    lib.ongym_step_policy.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.ongym_step_actions.argtypes = [vp, vp, vp]
    lib.ongym_policy_actions.argtypes = [vp, C.c_int32, vp, vp]
    if hasattr(lib, "ongym_step_actions_bundle"):
        lib.ongym_step_actions_bundle.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp, vp]
        lib.ongym_step_actions_bundle.restype = C.c_int32
    lib.ongym_observe.argtypes = [vp, vp, vp]
    if hasattr(lib, "ongym_sample_actions"):
        lib.ongym_sample_actions.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp]
        lib.ongym_sample_actions.restype = C.c_int32
    lib.ongym_query_available.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.ongym_query_gsnr.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ongym_query_gsnr_many.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    lib.ongym_query_moves.argtypes = [vp, C.c_int32, vp, vp]
    lib.ongym_query_grid.argtypes = [vp, C.c_int32, vp]
    lib.ongym_query_candidates.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.ongym_query_path_free.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ongym_query_services.argtypes = [vp, C.c_int32, vp, vp]
    lib.ongym_query_request.argtypes = [vp, C.c_int32, vp]
    lib.ongym_stats_get.argtypes = [vp, vp]
    lib.ongym_query_occupancy.argtypes = [vp, vp, vp, vp]
    if os.environ.get("ONGYM_HIP_LIB") and not hasattr(lib, "ongym_query_occupancy_policy"):
        # an older experiment build named by ONGYM_HIP_LIB (tools/ab_bench.py): first fit only
        lib.ongym_query_occupancy_policy = lambda h, policy, nb, lds, lean: lib.ongym_query_occupancy(h, nb, lds, lean)
        return _declare_tail(lib, vp, skip=("ongym_query_occupancy_policy",))
    lib.ongym_query_occupancy_policy.argtypes = [vp, C.c_int32, vp, vp, vp]
    _declare_tail(lib, vp)


def _declare_tail(lib, vp, skip=()):
    lib.ongym_sync.argtypes = [vp]
    if hasattr(lib, "ongym_set_stream"):
        lib.ongym_set_stream.argtypes = [vp, vp, C.c_int32]
        lib.ongym_set_stream.restype = C.c_int32
    lib.ongym_last_kernel_ms.argtypes = [vp]
    lib.ongym_last_kernel_ms.restype = C.c_double
    lib.ongym_last_error.argtypes = [vp]
    lib.ongym_last_error.restype = C.c_char_p
    lib.ongym_abi_version.argtypes = []
    lib.ongym_sizeof.argtypes = [C.c_int32]
    for name in ("ongym_create", "ongym_seed", "ongym_seed_base", "ongym_set_requests", "ongym_reset", "ongym_reset_episode_counters", "ongym_step_policy",
                 "ongym_step_actions", "ongym_policy_actions", "ongym_observe", "ongym_query_available", "ongym_query_gsnr", "ongym_query_gsnr_many", "ongym_query_moves",
                 "ongym_query_grid", "ongym_query_services", "ongym_query_request", "ongym_stats_get", "ongym_sync",
                 "ongym_abi_version", "ongym_sizeof", "ongym_query_candidates", "ongym_query_path_free", "ongym_observe", "ongym_query_occupancy", "ongym_query_occupancy_policy"):
        if name not in skip:
            getattr(lib, name).restype = C.c_int32


EXPORTED_SYMBOLS = (
    "ongym_create", "ongym_destroy", "ongym_seed", "ongym_seed_base", "ongym_set_requests", "ongym_reset", "ongym_reset_episode_counters", "ongym_step_policy",
    "ongym_step_actions", "ongym_step_actions_bundle", "ongym_policy_actions", "ongym_observe", "ongym_sample_actions", "ongym_query_available", "ongym_query_gsnr", "ongym_query_gsnr_many", "ongym_query_moves", "ongym_query_grid",
    "ongym_query_services", "ongym_query_request", "ongym_query_candidates", "ongym_query_path_free",
    "ongym_stats_get", "ongym_sync", "ongym_set_stream", "ongym_last_kernel_ms", "ongym_query_occupancy", "ongym_query_occupancy_policy",
    "ongym_last_error", "ongym_abi_version", "ongym_sizeof")


def load_library(path: Optional[str] = None):
    """dlopen libongym_hip.so (built in-tree by __graft_entry__.build()). Raises if it is missing: there is no
    fallback implementation."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or os.environ.get("ONGYM_HIP_LIB") or HIP_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build the HIP extension first "
                           f"(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = C.CDLL(path)
    _declare(lib)
    if lib.ongym_abi_version() != ABI_VERSION:
        raise RuntimeError("libongym_hip.so ABI version mismatch")
    sizes = (C.sizeof(OngymConfig), REQUEST_DTYPE.itemsize, STEP_DTYPE.itemsize, SERVICE_DTYPE.itemsize,
             STATS_DTYPE.itemsize, MOVE_DTYPE.itemsize)
    for what, expect in enumerate(sizes):
        got = lib.ongym_sizeof(what)
        if got != expect:
            raise RuntimeError(f"struct size mismatch for #{what}: library {got}, python {expect}")
    _LIB = lib
    return lib
