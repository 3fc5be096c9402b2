"""Gymnasium adapter with the reference's surface (optical_networking_gym/wrappers/qrmsa_gym.py:24-87)."""
from __future__ import annotations

from typing import Any

import numpy as np

from ..envs.qrmsa import QRMSAEnv

try:
    import gymnasium as gym
    from gymnasium.envs.registration import register
    _Base = gym.Env
    try:
        register(id="QRMSAEnvWrapper-v0", entry_point="optical_networking_gym.wrappers.qrmsa_gym:QRMSAEnvWrapper")
    except Exception:  # already registered
        pass
except Exception:  # gymnasium absent: the adapter still works as a plain object
    gym = None
    _Base = object


class QRMSAEnvWrapper(_Base):
    metadata = {"render_modes": ["human"]}

    def __init__(self, *args, bands=None, **kwargs):
        super().__init__()
        if bands is not None:
            kwargs["bands"] = bands
        self.env = QRMSAEnv(*args, **kwargs)
        self.action_space = self.env.action_space
        self.observation_space = self.env.observation_space
        self.num_spectrum_resources = kwargs.get("num_spectrum_resources", 320)
        self.bit_rates = kwargs.get("bit_rates", (10, 40, 100))
        self.channel_width = kwargs.get("channel_width", 12.5)
        self.seed_value = kwargs.get("seed", 10)
        self._last_mask = None

    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        self._last_mask = info.get("mask", self._last_mask)
        return obs, info

    def step(self, action: Any):
        obs, reward, done, truncated, info = self.env.step(action)
        self._last_mask = info.get("mask", self._last_mask)
        return obs, reward, done, truncated, info

    def render(self, mode="human"):
        return None

    def close(self):
        return self.env.close()

    def action_masks(self):
        return self._last_mask

    def get_available_slots(self, route):
        return self.env.get_available_slots(route)

    def get_number_slots(self, service, modulation):
        return self.env.get_number_slots(service, modulation)

    def get_available_blocks(self, idp, num_slots, j):
        return self.env.get_available_blocks(idp, num_slots, j)
