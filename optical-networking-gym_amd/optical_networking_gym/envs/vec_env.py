"""VecEnv-shaped batched environment for RL loops (the role of SB3's `SubprocVecEnv` over 14 `QRMSAEnvWrapper`
processes in the reference's examples/ONDM_2025/train_multi_masked_ppo.py:410-412 — here ONE device environment).

Semantics follow stable-baselines3's VecEnv convention so a MaskablePPO-style loop can consume it:
* `reset()` -> obs float32 [B, obs_dim]
* `step(actions)` -> (obs, rewards float32 [B], dones bool [B], infos list[dict]); a replica that terminates is reset
  immediately: its returned obs is the first observation of the new episode and `infos[i]["terminal_observation"]`
  holds nothing costly (the reference's observation of a finished episode is never used by the agents) but the episode
  statistics snapshot is there (`infos[i]["episode"]`).
* `action_masks()` -> bool [B, n_actions]  (info['mask'] of the reference, wrappers/qrmsa_gym.py:74-75)
Invalid actions keep the reference's behaviour: occupied slots -> penalty reward, same request (quirk Q5); a
QoT-infeasible action (the reference raises ValueError, qrmsa.pyx:925-929) is reported as reward -3 and
`infos[i]["qot_error"] = True` without touching the network — masked agents never produce one.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .. import _native as nat
from .batched import BatchedQRMSAEnv


class QRMSAVecEnv:
    def __init__(self, topology=None, *, num_envs: int, seed: int = 0, **kwargs):
        kwargs.setdefault("auto_reset", True)
        self.env = BatchedQRMSAEnv(topology, batch_size=num_envs, **kwargs)
        self.num_envs = int(num_envs)
        c = self.env.holder.struct
        self.obs_dim = 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12
        self.n_actions = c.k_paths * c.n_mods_consider * c.n_slots + 1
        self.env.seed(seed)
        self._mask = None
        self._obs = None

    def _observe(self):
        self._obs, mask = self.env.observe()
        self._mask = mask.astype(bool)
        return self._obs

    def reset(self) -> np.ndarray:
        self.env.reset()
        return self._observe()

    def action_masks(self) -> np.ndarray:
        if self._mask is None:
            self._observe()
        return self._mask

    def step(self, actions: Sequence[int]):
        rec = self.env.step(np.asarray(actions, np.int32))
        rewards = rec["reward"].astype(np.float32)
        qot = (rec["flags"] & nat.F_QOT_ERROR) != 0
        rewards[qot] = -3.0
        dones = rec["terminated"].astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        if dones.any() or qot.any():
            st = self.env.stats() if dones.any() else None
            for i in np.flatnonzero(dones | qot):
                if qot[i]:
                    infos[i]["qot_error"] = True
                if dones[i]:
                    s = st[i]
                    infos[i]["episode"] = {
                        "episode_service_blocking_rate": float(s["last_episode_service_blocking_rate"]),
                        "episode_bit_rate_blocking_rate": float(s["last_episode_bit_rate_blocking_rate"]),
                        "episode_services_accepted": int(s["last_episode_accepted"]),
                        "mean_gsnr": float(s["last_mean_gsnr"]),
                    }
        return self._observe(), rewards, dones, infos

    def close(self):
        self.env.close()
