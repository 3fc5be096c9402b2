"""Batched QRMSA environment: B independent replicas stepped by the HIP kernels behind include/ongym.h.

This is the host-side mirror of the reference's `QRMSAEnv` (envs/qrmsa.pyx:118-1644) for the hot path only:
constructor kwargs keep the reference's names (qrmsa.pyx:206-237) plus `batch_size`, `capacity`, `auto_reset`,
`device`; `reset()/step()` keep their meaning per replica; the first-fit heuristic
(heuristics/heuristics.py:923-966) is fused on device (`step_policy`).

The class is a thin ctypes shim: no arithmetic of the hot path happens in Python, and there is no CPU fallback —
constructing it without a built libongym_hip.so or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from .. import _native as nat
from .._tables import StaticTables


class OngymError(RuntimeError):
    pass


class BatchedQRMSAEnv:
    def __init__(self, topology=None, *, tables: Optional[StaticTables] = None, batch_size: int = 1,
                 modulations: Optional[Sequence] = None, capacity: int = 1024, auto_reset: bool = True,
                 device: int = 0, io_device: bool = False, **kwargs):
        if tables is None:
            if topology is None:
                raise ValueError("need a topology graph (from get_topology) or StaticTables")
            tables = StaticTables.from_topology(topology)
        if modulations is None:
            modulations = topology.graph.get("modulations") if topology is not None else None
        if not modulations:
            raise ValueError("no modulations: pass `modulations=` or build the topology with them")
        modulations = list(modulations)
        # modulations_to_consider < len(modulations) (qrmsa.pyx:313): the action space addresses a window of that many formats
        # below max_modulation_idx (codec :801-834); observation() moves the window per request (:543-581, 712-717)
        mtc = min(int(kwargs.pop("modulations_to_consider", len(modulations))), len(modulations))
        kwargs["modulations_to_consider"] = mtc
        for dead in ("seed", "allow_rejection", "reset", "file_name", "blocks_to_consider", "gen_observation",
                     "bands", "bandwidth", "k_paths"):
            kwargs.pop(dead, None)
        self.holder = nat.ConfigHolder(tables, modulations=modulations, batch=batch_size, capacity=capacity,
                                       auto_reset=auto_reset, device=device, io_device=io_device, **kwargs)
        self.tables = tables
        self.modulations = modulations
        self.batch_size = int(batch_size)
        self.lib = nat.load_library()
        h = C.c_void_p()
        rc = self.lib.ongym_create(C.byref(self.holder.struct), C.byref(h))
        if rc != 0:
            raise OngymError(f"ongym_create failed ({rc}): {self.lib.ongym_last_error(None).decode()}")
        self._h = h
        self._trace = None

    # ------------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.ongym_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise OngymError(f"{what} failed ({rc}): {self.lib.ongym_last_error(self._h).decode()}")

    @property
    def reject_action(self) -> int:
        return self.holder.reject_action

    @property
    def num_actions(self) -> int:
        return self.holder.reject_action + 1

    # ---- request sources -----------------------------------------------------------------------------------------
    def seed(self, seed: int, replica_base: int = 0):
        """Device traffic generator; local replica r uses stream (seed, replica_base + r) of include/ongym_traffic.h.
        `replica_base` = index of this environment's first replica in a batch sharded over several environments."""
        self.replica_base = int(replica_base)
        self._check(self.lib.ongym_seed_base(self._h, C.c_uint64(seed), C.c_uint64(replica_base)), "ongym_seed_base")

    def set_requests(self, requests: np.ndarray):
        """Trace replay: `requests` is a REQUEST_DTYPE array [batch, n] (or [n] for batch 1)."""
        req = np.ascontiguousarray(requests, nat.REQUEST_DTYPE)
        if req.ndim == 1:
            req = req[None, :]
        if req.shape[0] != self.batch_size:
            raise ValueError("requests must have one row per replica")
        self._trace = req
        self._check(self.lib.ongym_set_requests(self._h, req.ctypes.data, req.shape[1]), "ongym_set_requests")

    # ---- reset / step -----------------------------------------------------------------------------------------------
    def reset(self, mask: Optional[np.ndarray] = None):
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            if mask.shape != (self.batch_size,):
                raise ValueError("mask must have one entry per replica")
        self._check(self.lib.ongym_reset(self._h, mask.ctypes.data if mask is not None else None), "ongym_reset")

    def reset_episode_counters(self, mask: Optional[np.ndarray] = None):
        """`reset(options={"only_episode_counters": True})` of the reference (qrmsa.pyx:427-464) per replica: episode counters
        and histograms to zero, the departure heap dropped (running services stay for good), nothing else touched."""
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            if mask.shape != (self.batch_size,):
                raise ValueError("mask must have one entry per replica")
        self._check(self.lib.ongym_reset_episode_counters(self._h, mask.ctypes.data if mask is not None else None),
                    "ongym_reset_episode_counters")

    def step_policy(self, nsteps: int = 1, record: bool = True, policy: int = nat.POLICY_FIRST_FIT,
                    out_device_ptr: Optional[int] = None):
        """`nsteps` x {heuristic `policy`; step}. Returns STEP_DTYPE [nsteps, batch] when `record`, else None.  With an
        environment created with io_device=True, `out_device_ptr` is the address of a device buffer of nsteps * batch step
        records (e.g. torch.Tensor.data_ptr()): the records stay on the device and the call returns without synchronising."""
        if out_device_ptr is not None:
            if not self.holder.struct.io_device:
                raise ValueError("out_device_ptr needs an environment created with io_device=True")
            self._check(self.lib.ongym_step_policy(self._h, policy, nsteps, C.c_void_p(out_device_ptr)), "ongym_step_policy")
            return None
        out = np.zeros((nsteps, self.batch_size), nat.STEP_DTYPE) if record else None
        self._check(self.lib.ongym_step_policy(self._h, policy, nsteps, out.ctypes.data if record else None),
                    "ongym_step_policy")
        return out

    def step(self, actions: np.ndarray) -> np.ndarray:
        actions = np.ascontiguousarray(actions, np.int32)
        if actions.shape != (self.batch_size,):
            raise ValueError("actions must have one entry per replica")
        out = np.zeros(self.batch_size, nat.STEP_DTYPE)
        self._check(self.lib.ongym_step_actions(self._h, actions.ctypes.data, out.ctypes.data), "ongym_step_actions")
        return out

    def step_bundle(self, actions: np.ndarray, next_policy: int = -1):
        """`step(actions)`, then fused policy `next_policy` on the new current requests (skipped when negative), everything
        behind one synchronisation (ongym_step_actions_bundle): (records, requests, stats, next_actions | None, next_flags | None)."""
        actions = np.ascontiguousarray(actions, np.int32)
        if actions.shape != (self.batch_size,):
            raise ValueError("actions must have one entry per replica")
        B = self.batch_size
        rec, req, st = np.zeros(B, nat.STEP_DTYPE), np.zeros(B, nat.REQUEST_DTYPE), np.zeros(B, nat.STATS_DTYPE)
        na, nf = (np.zeros(B, np.int32), np.zeros(B, np.uint8)) if next_policy >= 0 else (None, None)
        self._check(self.lib.ongym_step_actions_bundle(self._h, actions.ctypes.data, int(next_policy), rec.ctypes.data,
                                                       req.ctypes.data, st.ctypes.data,
                                                       na.ctypes.data if na is not None else None,
                                                       nf.ctypes.data if nf is not None else None), "ongym_step_actions_bundle")
        return rec, req, st, na, nf

    def policy_actions(self, policy: int = nat.POLICY_FIRST_FIT):
        actions = np.zeros(self.batch_size, np.int32)
        flags = np.zeros(self.batch_size, np.uint8)
        self._check(self.lib.ongym_policy_actions(self._h, policy, actions.ctypes.data, flags.ctypes.data),
                    "ongym_policy_actions")
        return actions, flags

    def sample_actions(self, mask: np.ndarray, seed: int, draw_index: int) -> np.ndarray:
        """One uniformly random valid action per replica from an action mask [batch, n_actions] (gymnasium's
        `action_space.sample(mask=...)` on the reference's Discrete space), drawn on device (ongym_sample_actions)."""
        mask = np.ascontiguousarray(mask, np.uint8)
        if mask.shape != (self.batch_size, self.num_actions):
            raise ValueError("mask must be [batch, n_actions]")
        actions = np.zeros(self.batch_size, np.int32)
        self._check(self.lib.ongym_sample_actions(self._h, mask.ctypes.data, C.c_uint64(seed), C.c_uint64(draw_index),
                                                  actions.ctypes.data), "ongym_sample_actions")
        return actions

    def observe(self):
        """observation() + action mask of every replica's current request: (float32 [B, obs_dim], uint8 [B, n_actions])."""
        c = self.holder.struct
        obs = np.zeros((self.batch_size, 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12), np.float32)
        mask = np.zeros((self.batch_size, c.k_paths * c.n_mods_consider * c.n_slots + 1), np.uint8)
        self._check(self.lib.ongym_observe(self._h, obs.ctypes.data, mask.ctypes.data), "ongym_observe")
        return obs, mask

    # ---- queries (plugin API) ----------------------------------------------------------------------------------------
    def available_slots(self, replica: int, path_id: int) -> np.ndarray:
        out = np.zeros(self.holder.struct.n_slots, np.int32)
        self._check(self.lib.ongym_query_available(self._h, replica, path_id, out.ctypes.data), "ongym_query_available")
        return out

    def gsnr(self, replica: int, path_id: int, slot: int, nslots: int) -> np.ndarray:
        out = np.zeros(3, np.float64)
        self._check(self.lib.ongym_query_gsnr(self._h, replica, path_id, slot, nslots, out.ctypes.data),
                    "ongym_query_gsnr")
        return out

    def gsnr_many(self, replica: int, cands) -> np.ndarray:
        """`calculate_osnr` for many candidates `(path_id, slot, nslots)` of one replica in one launch -> [count][3] dB."""
        cands = np.ascontiguousarray(cands, np.int32).reshape(-1, 3)
        out = np.zeros((len(cands), 3), np.float64)
        self._check(self.lib.ongym_query_gsnr_many(self._h, replica, len(cands), cands.ctypes.data, out.ctypes.data),
                    "ongym_query_gsnr_many")
        return out

    def moves(self, replica: int):
        """Reallocations defragment() made during the replica's last step: (records, total count)."""
        out = np.zeros(nat.MOVE_LOG, nat.MOVE_DTYPE)
        n = C.c_int32(0)
        self._check(self.lib.ongym_query_moves(self._h, replica, out.ctypes.data, C.byref(n)), "ongym_query_moves")
        return out[:min(n.value, nat.MOVE_LOG)], n.value

    def candidates(self, row: np.ndarray, nslots: int) -> list:
        """`_get_candidates(row, nslots, len(row))` evaluated on device."""
        row = np.ascontiguousarray(row, np.int32)
        out = np.zeros(len(row), np.int32)
        n = C.c_int32(0)
        self._check(self.lib.ongym_query_candidates(self._h, row.ctypes.data, len(row), int(nslots), out.ctypes.data,
                                                    C.byref(n)), "ongym_query_candidates")
        return out[:n.value].tolist()

    def is_path_free(self, replica: int, path_id: int, slot: int, nslots: int) -> bool:
        out = C.c_int32(0)
        self._check(self.lib.ongym_query_path_free(self._h, replica, path_id, int(slot), int(nslots), C.byref(out)),
                    "ongym_query_path_free")
        return bool(out.value)

    def grid(self, replica: int) -> np.ndarray:
        c = self.holder.struct
        out = np.zeros((c.n_links, c.n_slots), np.int32)
        self._check(self.lib.ongym_query_grid(self._h, replica, out.ctypes.data), "ongym_query_grid")
        return out

    def services(self, replica: int) -> np.ndarray:
        out = np.zeros(self.holder.struct.capacity, nat.SERVICE_DTYPE)
        n = C.c_int32(0)
        self._check(self.lib.ongym_query_services(self._h, replica, out.ctypes.data, C.byref(n)),
                    "ongym_query_services")
        return out[:n.value]

    def request(self, replica: int):
        out = np.zeros(1, nat.REQUEST_DTYPE)
        self._check(self.lib.ongym_query_request(self._h, replica, out.ctypes.data), "ongym_query_request")
        return out[0]

    def stats(self) -> np.ndarray:
        out = np.zeros(self.batch_size, nat.STATS_DTYPE)
        self._check(self.lib.ongym_stats_get(self._h, out.ctypes.data), "ongym_stats_get")
        return out

    def occupancy(self, policy: int = nat.POLICY_FIRST_FIT) -> dict:
        """Resident replicas (wavefronts) per compute unit of the kernel step_policy(policy=...) launches, its LDS bytes per
        replica, and whether a lean kernel is the one that runs."""
        nb, lds, lean = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self.lib.ongym_query_occupancy_policy(self._h, int(policy), C.byref(nb), C.byref(lds), C.byref(lean)),
                    "ongym_query_occupancy_policy")
        return dict(blocks_per_cu=nb.value, lds_bytes=lds.value, lean_kernel=bool(lean.value))

    def set_stream(self, stream_handle: Optional[int]):
        """Run this environment's launches on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); None
        returns to the environment's own stream.  See ongym_set_stream (include/ongym.h)."""
        if stream_handle is None:
            self._check(self.lib.ongym_set_stream(self._h, None, 1), "ongym_set_stream")
        else:       # 0 is HIP's default (null) stream: torch's default current stream
            self._check(self.lib.ongym_set_stream(self._h, C.c_void_p(int(stream_handle)), 0), "ongym_set_stream")

    def sync(self):
        self._check(self.lib.ongym_sync(self._h), "ongym_sync")

    def last_kernel_ms(self) -> float:
        return float(self.lib.ongym_last_kernel_ms(self._h))
