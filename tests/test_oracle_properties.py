"""Property tests (hypothesis) of the oracle's primitives against definitions written from SURVEY Appendix A, on inputs the
captured known-answer cases do not enumerate: the candidate rule (A.2), the action codec round trip (A.5), slots needed (A.4)
and provision / release on the grid (A.3).  CPU only; the GPU suite holds the HIP kernels to the oracle."""
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from common import golden_tables, jocn_modulations
from optical_networking_gym._native import ConfigHolder
from oracle_lib import OracleEnv


@pytest.fixture(scope="module")
def env():
    h = ConfigHolder(golden_tables("nsfnet"), modulations=jocn_modulations(), num_spectrum_resources=320, batch=1,
                     capacity=1024, load=300, bit_rate_selection="discrete", bit_rates=(10, 40, 100, 400))
    return OracleEnv(h)


def candidates_by_definition(row, n):
    """A.2: start s is feasible iff s + n <= S and every slot of [s, min(s + n, S - 1)] is free (n slots + a right guard
    slot, the guard waived only when the allocation ends exactly at S)."""
    S = len(row)
    return [s for s in range(S) if s + n <= S and all(row[j] == 1 for j in range(s, min(s + n, S - 1) + 1))]


@settings(max_examples=300, deadline=None)
@given(st.lists(st.integers(0, 1), min_size=1, max_size=130), st.integers(1, 40), st.integers(0, 5))
def test_candidate_rule(env, bits, n, run_bias):
    row = np.array(bits, np.int32)
    if run_bias:                      # long runs: stretch every bit
        row = np.repeat(row, run_bias + 1)[:320]
    assert env.candidates(row, n) == candidates_by_definition(row.tolist(), n)


def test_candidate_rule_example_of_the_survey(env):
    assert env.candidates(np.array([1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1], np.int32), 2) == [0, 4, 5, 9, 10]   # A.2 [measured]


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 4), st.integers(0, 5), st.integers(0, 319))
def test_action_codec_round_trip(env, route, mod, slot):
    a = env.encode(route, mod, slot)
    assert 0 <= a < env.reject_action
    assert list(env.decode(a)) == [route, mod, slot]


@settings(max_examples=200, deadline=None)
@given(st.floats(1.0, 2000.0, allow_nan=False), st.integers(0, 5))
def test_slots_needed(env, bit_rate, mod):
    se = [1, 2, 3, 4, 5, 6][mod]
    br = float(np.float32(bit_rate))              # Service.bit_rate is a C float (A.6)
    assert env.number_slots(br, mod) == math.ceil(br / (se * 12.5))   # A.4


def test_provision_marks_the_guard_slot_and_release_frees_it(env):
    """A.3 on a fresh network through the public step: accept first fit, check the marked block incl. the guard."""
    env.seed(3); env.reset()
    act, _, _ = env.policy_first_fit()
    route, mod, slot = env.decode(act)
    rc, rec = env.step(act)
    assert rc == 0 and rec["accepted"] and rec["slot"] == slot == 0
    n = int(rec["nslots"])
    grid = env.grid()
    links = [l for l in range(grid.shape[0]) if grid[l, 0] == 0]
    assert links and all((grid[l, :n + 1] == 0).all() and (grid[l, n + 1:] == 1).all() for l in links)
