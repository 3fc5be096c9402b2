"""Gymnasium-facing adapter around the single-replica compatibility env.

Surface kept from the reference (optical_networking_gym/wrappers/qrmsa_gym.py:24-87): constructor kwargs are
`QRMSAEnv`'s, `reset(*, seed, options)`, `step(action)`, `action_masks()` for MaskablePPO, `.env` for the heuristics'
unwrap chain, and the helper pass-throughs the example scripts call on the wrapper. Everything is delegated to the
device-backed `QRMSAEnv`; the adapter only remembers the last action mask.
"""
from __future__ import annotations

from ..envs.qrmsa import QRMSAEnv

try:
    import gymnasium as _gymnasium
    from gymnasium.envs.registration import register as _register
    _EnvBase = _gymnasium.Env
except Exception:  # gymnasium is optional: the adapter is then a plain object with the same methods
    _gymnasium, _register, _EnvBase = None, None, object

ENV_ID = "QRMSAEnvWrapper-v0"
if _register is not None:
    try:
        _register(id=ENV_ID, entry_point=f"{__name__}:QRMSAEnvWrapper")
    except Exception:
        pass

# helper calls that scripts make on the wrapper instead of on `.env`
_FORWARDED = frozenset({"get_available_slots", "get_number_slots", "get_available_blocks"})


class QRMSAEnvWrapper(_EnvBase):
    metadata = {"render_modes": ["human"]}

    def __init__(self, *args, bands=None, **kwargs):
        super().__init__()
        if bands is not None:
            kwargs["bands"] = bands
        inner = QRMSAEnv(*args, **kwargs)
        self.env = inner
        self.action_space, self.observation_space = inner.action_space, inner.observation_space
        self.num_spectrum_resources = kwargs.get("num_spectrum_resources", 320)
        self.bit_rates = kwargs.get("bit_rates", (10, 40, 100))
        self.channel_width = kwargs.get("channel_width", 12.5)
        self.seed_value = kwargs.get("seed", 10)
        self._mask = None

    def _remember(self, info):
        self._mask = info.get("mask", self._mask)

    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        self._remember(info)
        return obs, info

    def step(self, action):
        result = self.env.step(action)
        self._remember(result[4])
        return result

    def action_masks(self):
        return self._mask

    def __getattr__(self, name):
        if name in _FORWARDED:
            return getattr(self.env, name)
        raise AttributeError(name)

    def render(self, mode="human"):
        return None

    def close(self):
        return self.env.close()
