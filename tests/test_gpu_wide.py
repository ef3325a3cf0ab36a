"""The row-wide Fp arithmetic (csrc/wide.cuh: one field element across the 16 lanes of a DPP row) against plain integer
arithmetic: the single-verification latency path is built on it."""
import random

import pytest

import util
from util import P

pytestmark = pytest.mark.gpu


def test_wide_multiplier_matches_integers(api):
    rng = random.Random(77)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2 ** 380, 2 ** 381 % P, (1 << 28) - 1, 1 << 28, (1 << 364)]
    vals = edge + [rng.randrange(P) for _ in range(500)]
    a = vals
    b = [vals[(7 * i + 3) % len(vals)] for i in range(len(vals))]
    got = api.debug_wide_mul([util.fp_raw(x) for x in a], [util.fp_raw(y) for y in b])
    for x, y, g in zip(a, b, got):
        assert util.fp_from_raw(g) == x * y % P
    # dependent chains: a * b^reps (the lazy limbs of one product feed the next)
    for reps in (2, 17, 300):
        got = api.debug_wide_mul([util.fp_raw(x) for x in a[:40]], [util.fp_raw(y) for y in b[:40]], reps)
        for x, y, g in zip(a, b, got):
            assert util.fp_from_raw(g) == x * pow(y, reps, P) % P
