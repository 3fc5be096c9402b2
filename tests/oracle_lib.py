"""ctypes binding of oracle/libongym_oracle.so — TEST INFRASTRUCTURE (the checker), never imported by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from optical_networking_gym._native import (ConfigHolder, OngymConfig, REQUEST_DTYPE, SERVICE_DTYPE, STATS_DTYPE,
                                            STEP_DTYPE)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.environ.get("ONGYM_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libongym_oracle.so")   # override: sanitizer builds

_lib = None


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "ongym_oracle.c")
    stale = (not os.path.exists(ORACLE_LIB)) or any(
        os.path.getmtime(p) > os.path.getmtime(ORACLE_LIB)
        for p in (src, os.path.join(REPO, "include", "ongym.h"), os.path.join(REPO, "include", "ongym_traffic.h")))
    if os.environ.get("ONGYM_ORACLE_LIB"):
        return ORACLE_LIB
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B", "libongym_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_LIB)
        vp = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OngymConfig), C.c_int]
        L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_seed.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.orc_set_trace.argtypes = [vp, vp, C.c_int64]
        L.orc_reset.argtypes = [vp]
        L.orc_reset_counters.argtypes = [vp]
        L.orc_reset_counters.restype = None
        L.orc_number_slots.argtypes = [vp, C.c_float, C.c_int]
        L.orc_available.argtypes = [vp, C.c_int, vp]
        L.orc_candidates.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.orc_is_path_free.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_gn.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_gn_lists.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.orc_encode_action.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_decode_action.argtypes = [vp, C.c_int, vp]
        L.orc_reject_action.argtypes = [vp]
        L.orc_policy_first_fit.argtypes = [vp, vp, vp]
        L.orc_policy.argtypes = [vp, C.c_int, vp, vp]
        L.orc_run_policy.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_step.argtypes = [vp, C.c_int, vp]
        L.orc_stats.argtypes = [vp, vp]
        L.orc_grid.argtypes = [vp, vp]
        L.orc_request.argtypes = [vp, vp]
        L.orc_services.argtypes = [vp, vp]
        L.orc_observe.argtypes = [vp, vp, C.c_double, vp, vp]
        L.orc_max_modulation_idx.argtypes = [vp]
        L.orc_run_first_fit.argtypes = [vp, C.c_int, vp]
        L.orc_batch_run_first_fit.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_batch_run_first_fit.restype = C.c_int64
        L.orc_batch_run_policy.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_batch_run_policy.restype = C.c_int64
        _lib = L
    return _lib


class OracleEnv:
    """One replica of the CPU restatement."""

    def __init__(self, holder: ConfigHolder, replica: int = 0):
        self.holder = holder
        self.cfg = holder.struct
        self.L = lib()
        self.h = self.L.orc_create(C.byref(self.cfg), replica)
        self.replica = replica
        self._trace = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def seed(self, seed: int):
        self.L.orc_seed(self.h, seed, self.replica)

    def set_trace(self, reqs: np.ndarray):
        self._trace = np.ascontiguousarray(reqs, REQUEST_DTYPE)
        self.L.orc_set_trace(self.h, self._trace.ctypes.data, len(self._trace))

    def reset(self):
        return self.L.orc_reset(self.h)

    def reset_counters(self):
        self.L.orc_reset_counters(self.h)

    def number_slots(self, bit_rate: float, mod: int) -> int:
        return self.L.orc_number_slots(self.h, bit_rate, mod)

    def available(self, path_id: int) -> np.ndarray:
        out = np.zeros(self.cfg.n_slots, np.int32)
        self.L.orc_available(self.h, path_id, out.ctypes.data)
        return out

    def candidates(self, row: np.ndarray, n: int) -> list:
        row = np.ascontiguousarray(row, np.int32)
        out = np.zeros(len(row) + 1, np.int32)
        cnt = self.L.orc_candidates(row.ctypes.data, len(row), n, out.ctypes.data, len(out))
        return out[:cnt].tolist()

    def is_path_free(self, path_id, slot, n) -> bool:
        return bool(self.L.orc_is_path_free(self.h, path_id, slot, n))

    def gn(self, path_id, slot, n) -> np.ndarray:
        out = np.zeros(3)
        self.L.orc_gn(self.h, path_id, slot, n, out.ctypes.data)
        return out

    def gn_lists(self, path_id, slot, n, counts, intf) -> np.ndarray:
        counts = np.ascontiguousarray(counts, np.int32)
        intf = np.ascontiguousarray(intf, np.int16).reshape(-1, 3)
        out = np.zeros(3)
        self.L.orc_gn_lists(self.h, path_id, slot, n, counts.ctypes.data, intf.ctypes.data, out.ctypes.data)
        return out

    def encode(self, path_index, mod, slot) -> int:
        return self.L.orc_encode_action(self.h, path_index, mod, slot)

    def decode(self, action) -> list:
        out = np.zeros(3, np.int32)
        self.L.orc_decode_action(self.h, action, out.ctypes.data)
        return out.tolist()

    @property
    def reject_action(self) -> int:
        return self.L.orc_reject_action(self.h)

    def policy_first_fit(self):
        a, b = C.c_int(0), C.c_int(0)
        act = self.L.orc_policy_first_fit(self.h, C.byref(a), C.byref(b))
        return act, bool(a.value), bool(b.value)

    def policy(self, policy_id: int):
        a, b = C.c_int(0), C.c_int(0)
        act = self.L.orc_policy(self.h, policy_id, C.byref(a), C.byref(b))
        return act, bool(a.value), bool(b.value)

    def run_policy(self, policy_id: int, nsteps: int) -> np.ndarray:
        rec = np.zeros(nsteps, STEP_DTYPE)
        rc = self.L.orc_run_policy(self.h, policy_id, nsteps, rec.ctypes.data)
        if rc:
            raise RuntimeError(f"oracle run failed rc={rc}")
        return rec

    def step(self, action: int):
        rec = np.zeros(1, STEP_DTYPE)
        rc = self.L.orc_step(self.h, action, rec.ctypes.data)
        return rc, rec[0]

    def run_first_fit(self, nsteps: int) -> np.ndarray:
        rec = np.zeros(nsteps, STEP_DTYPE)
        rc = self.L.orc_run_first_fit(self.h, nsteps, rec.ctypes.data)
        if rc:
            raise RuntimeError(f"oracle run failed rc={rc}")
        return rec

    def stats(self):
        s = np.zeros(1, STATS_DTYPE)
        self.L.orc_stats(self.h, s.ctypes.data)
        return s[0]

    def grid(self) -> np.ndarray:
        out = np.zeros((self.cfg.n_links, self.cfg.n_slots), np.int32)
        self.L.orc_grid(self.h, out.ctypes.data)
        return out

    def request(self):
        q = np.zeros(1, REQUEST_DTYPE)
        self.L.orc_request(self.h, q.ctypes.data)
        return q[0]

    def observe(self, path_len_norm: np.ndarray, max_bit_rate: float):
        c = self.cfg
        obs = np.zeros(1 + 2 + c.k_paths + c.k_paths * c.n_mods_consider * 12, np.float32)
        mask = np.zeros(c.k_paths * c.n_mods_consider * c.n_slots + 1, np.uint8)
        pl = np.ascontiguousarray(path_len_norm, np.float64)
        self.L.orc_observe(self.h, pl.ctypes.data, float(max_bit_rate), obs.ctypes.data, mask.ctypes.data)
        return obs, mask

    @property
    def max_modulation_idx(self) -> int:
        return self.L.orc_max_modulation_idx(self.h)

    def services(self) -> np.ndarray:
        out = np.zeros(self.cfg.capacity, SERVICE_DTYPE)
        n = self.L.orc_services(self.h, out.ctypes.data)
        return out[:n]


def batch_run_first_fit(envs, nsteps: int, threads: int) -> int:
    arr = (C.c_void_p * len(envs))(*[e.h for e in envs])
    return lib().orc_batch_run_first_fit(arr, len(envs), nsteps, threads)


def batch_run_policy(envs, policy: int, nsteps: int, threads: int) -> int:
    arr = (C.c_void_p * len(envs))(*[e.h for e in envs])
    return lib().orc_batch_run_policy(arr, len(envs), policy, nsteps, threads)
