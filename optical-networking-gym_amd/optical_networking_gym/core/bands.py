"""Transmission-band value objects (reference: optical_networking_gym/core/bands.py:1-20).

Only `bands[1]` (the C band) is ever consulted by the reference's QRMSAEnv, and only by `get_number_slots`, which then
divides by a channel width in Hz instead of GHz so that every service needs ONE slot (quirk Q9, qrmsa.pyx:1198-1205,
422-423). The compat env reproduces exactly that.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Band:
    name: str
    freq_start: float   # THz
    freq_end: float     # THz
    num_slots: int
    noise_figure: float
    attenuation: float
    input_power: float


def BandS() -> Band:
    return Band("S", 197.22, 205.30, 647, 7.0, 0.220, -0.38)


def BandC() -> Band:
    return Band("C", 191.60, 195.90, 344, 5.5, 0.191, -3.66)


def BandL() -> Band:
    return Band("L", 185.83, 190.90, 406, 6.0, 0.200, -2.78)
