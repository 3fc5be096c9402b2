"""MI355X-native batched QRMSA environment with the module layout of LEA-UFPA/optical-networking-gym.

Module paths mirror the reference so that its callers resolve unchanged
(`optical_networking_gym.topology`, `.envs.qrmsa`, `.wrappers.qrmsa_gym`, `.heuristics.heuristics`, `.core.osnr`,
`.utils`); the per-request hot path runs in hand-written HIP kernels behind the C ABI of include/ongym.h.
"""
__version__ = "0.1.0"
