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


def _f12_raw(f):
    return [util.fp_raw(x) for co in f for x in co]


def _f12_from(raws):
    v = [util.fp_from_raw(r) for r in raws]
    return tuple((v[2 * k], v[2 * k + 1]) for k in range(6))


def test_engine_operations_match_the_oracle(api):
    """every Fp12 operation table of the row-wide engine (csrc/wide_tables.cuh) on the device, one step at a time and in
    chains (unreduced product columns summed, one Montgomery reduction per output value), against the oracle's tower arithmetic.
    blsgpu_debug_wide_program starts with F = U = W = ACC = the input and returns T."""
    from oracle.py import bls381 as c
    rng = random.Random(5)
    for trial in range(3):
        a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
        if trial == 2:
            a = ((P - 1, P - 1),) * 6                                  # the largest canonical components
        ra = _f12_raw(a)
        run = lambda steps, reps=1: _f12_from(api.debug_wide_program(steps, ra, reps))  # noqa: E731
        assert run([('MUL', 'T', 'F', 'U')]) == c.f12_mul(a, a)
        assert run([('SQR', 'T', 'F', 'F')]) == c.f12_sqr(a)
        assert run([('FROB1', 'T', 'F', 'F')]) == c.f12_frob(a, 1)
        assert run([('FROB2', 'T', 'F', 'F')]) == c.f12_frob(a, 2)
        assert run([('CONJ', 'T', 'F', 'F')]) == c.f12_conj(a)
        assert run([('COPY', 'T', 'F', 'F')]) == a
        line = ((a[0], (0, 0), a[1], a[2], (0, 0), (0, 0)))           # U's first three coefficients as l0, l2 w^2, l3 w^3
        assert run([('MUL_LINE', 'T', 'F', 'U')]) == c.f12_mul(a, line)
        # aliased destination and a chain: ACC <- ACC^2 * F, four times
        want = a
        for _ in range(4):
            want = c.f12_mul(c.f12_sqr(want), a)
        assert run([('SQR', 'ACC', 'ACC', 'ACC'), ('MUL', 'ACC', 'ACC', 'F')] * 4 + [('COPY', 'T', 'ACC', 'ACC')]) == want
        # the cyclotomic squaring on an element of the cyclotomic subgroup: g = a^((p^6 - 1)(p^2 + 1)) built on the device
        t = c.f12_mul(c.f12_conj(a), c.f12_inv(a))
        g = c.f12_mul(c.f12_frob(t, 2), t)
        rg = _f12_raw(g)
        got = _f12_from(api.debug_wide_program([('CYC_SQR', 'ACC', 'ACC', 'ACC')] * 5 + [('COPY', 'T', 'ACC', 'ACC')], rg))
        want = g
        for _ in range(5):
            want = c.f12_sqr(want)
        assert got == want
        # the conjugating variant reads F and writes T directly (first / last step of a power by x)
        assert _f12_from(api.debug_wide_program([('CYC_SQR', 'ACC', 'F', 'F'), ('CYC_SQRC', 'T', 'ACC', 'ACC')], rg)) == c.f12_conj(c.f12_sqr(c.f12_sqr(g)))
    # the same program run repeatedly on its own output (reps): T <- T * F after T <- F
    a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
    got = _f12_from(api.debug_wide_program([('MUL', 'W', 'W', 'F'), ('COPY', 'T', 'W', 'W')], _f12_raw(a), 6))
    assert got == c.f12_pow(a, 7)
    with pytest.raises(api.BlsGpuRuntimeError):
        api.debug_wide_program([('PDBL1', 'T', 'F', 'F')], _f12_raw(a))   # not an Fp12 operation: refused on the host


def test_engine_fp12_inversion_program(api):
    """the Fp12 inversion of the final exponentiation's easy part as the engine runs it: table operations (norm to Fp6, to Fp2,
    to Fp) around the interpreter's one built-in, an inversion in Fp on a lone lane (variable-time safegcd)"""
    from oracle.py import bls381 as c
    rng = random.Random(9)
    inv_prog = [('CONJ', 'U', 'F', 'F'), ('MUL', 'T', 'F', 'U'), ('F6INV1', 'T', 'T', 'T'), ('F6INV2', 'W', 'T', 'T'), ('F6INV3', 'W', 'W', 'W'),
                ('FPINV', ('W', 3), ('W', 2), ('W', 2)), ('F6INV4', 'W', 'W', 'W'), ('F6INV5', 'T', 'T', 'W'), ('MUL', 'T', 'U', 'T')]
    for trial in range(4):
        a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
        if trial == 3:
            a = ((5, 0),) + ((0, 0),) * 5                  # an element of the prime field
        got = _f12_from(api.debug_wide_program(inv_prog, _f12_raw(a)))
        assert got == c.f12_inv(a)
        assert c.f12_mul(got, a) == c.F12_ONE
    # the built-in alone, on single values: T[0] <- F[0]^-1, T[1] <- F[3]^-1
    a = tuple((rng.randrange(1, P), rng.randrange(1, P)) for _ in range(6))
    got = api.debug_wide_program([('FPINV', ('T', 0), ('F', 0), ('F', 0)), ('FPINV', ('T', 1), ('F', 3), ('F', 3))], _f12_raw(a))
    assert util.fp_from_raw(got[0]) == pow(a[0][0], -1, P) and util.fp_from_raw(got[1]) == pow(a[1][1], -1, P)
