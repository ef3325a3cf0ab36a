"""Integer simulation of the row-wide engine's semantics (csrc/wide_engine.cuh) over the operation tables and programs that
tools/gen_wide_tables.py builds, checked against the oracle's arithmetic (oracle/py/bls381.py): every Fp12 operation, the hard
part of the final exponentiation, whole pairing checks (valid and invalid) uncut and cut into PRE_LINES / PRE_F1 / PRE_F1G / POST,
the sixteen-point sum programs of both groups and the cofactor clearing of hash-to-G2.  Test infrastructure (tests/test_wide_tables.py)."""
import importlib.util
import os
import random

import util
from oracle.py import bls381 as c

_spec = importlib.util.spec_from_file_location('gen_wide_tables', os.path.join(util.ROOT, 'tools', 'gen_wide_tables.py'))
g = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(g)
P = g.P
assert P == c.P
DST, SA, SB, TMP, CONST = g.DST, g.SA, g.SB, g.TMP, g.CONST
OPS, OPS_PT, PROGRAMS, PROGRAMS_PT = g.OPS, g.OPS_PT, g.PROGRAMS, g.PROGRAMS_PT
NSTEPS, PT_POINTS, G1S, G2S = g.NSTEPS, g.PT_POINTS, g.G1S, g.G2S
layout_f12, layout_pt, prog_final_hard, prog_key_lines = g.layout_f12, g.layout_pt, g.prog_final_hard, g.prog_key_lines
g1_point_off, g2_point_off = g.g1_point_off, g.g2_point_off
TABLES = g       # (functions below use g and gen as local names)


def consts24():
    out = []
    for j in (1, 2):
        for k in range(6):
            out += list(c.f2_pow(c.XI, k * (P ** j - 1) // 6))
    return out


def exec_op(op, dst, a, b):
    """one engine step on integer arrays (lists, modified in place; dst may be a or b).  Products first, then the linear rows
    ONE BY ONE in table order: a row must not read what an earlier row wrote -- exactly the device's freedom."""
    arrays = {DST: dst, SA: a, SB: b, CONST: consts24()}
    tmp = [0] * (op.ntmp + 1)
    arrays[TMP] = tmp
    V = lambda i: arrays[i >> 12][i & 0xfff]  # noqa: E731
    for x, y, out in op.prods:
        tmp[out & 0xfff] = (sum(k * V(i) for k, i in x) % P) * (sum(k * V(i) for k, i in y) % P) % P
    for terms, out in op.lins:
        dst[out & 0xfff] = sum(k * V(i) for k, i in terms) % P


def run(op, a12, b12=None, alias=False):
    a = list(a12)
    b = list(b12) if b12 is not None else a
    dst = a if alias else [0] * 12
    exec_op(op, dst, a, b)
    return dst


def flat(f):
    return [x for co in f for x in co]


def unflat(v):
    return tuple((v[2 * k], v[2 * k + 1]) for k in range(6))


def self_check():
    rng = random.Random(1)
    rnd12 = lambda: tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))  # noqa: E731
    by = {o.name: o for o in OPS}
    for _ in range(3):
        a, b = rnd12(), rnd12()
        assert unflat(run(by['MUL'], flat(a), flat(b))) == c.f12_mul(a, b)
        assert unflat(run(by['MUL'], flat(a), flat(b), alias=True)) == c.f12_mul(a, b)
        assert unflat(run(by['SQR'], flat(a))) == c.f12_sqr(a)
        assert unflat(run(by['SQR'], flat(a), alias=True)) == c.f12_sqr(a)
        assert unflat(run(by['FROB1'], flat(a))) == c.f12_frob(a, 1)
        assert unflat(run(by['FROB2'], flat(a))) == c.f12_frob(a, 2)
        assert unflat(run(by['FROB1'], flat(a), alias=True)) == c.f12_frob(a, 1)
        assert unflat(run(by['CONJ'], flat(a), alias=True)) == c.f12_conj(a)
        assert unflat(run(by['COPY'], flat(a))) == a
        # a cyclotomic element: f^((p^6 - 1)(p^2 + 1))
        t = c.f12_mul(c.f12_conj(a), c.f12_inv(a))
        g = c.f12_mul(c.f12_frob(t, 2), t)
        assert unflat(run(by['CYC_SQR'], flat(g))) == c.f12_sqr(g)
        assert unflat(run(by['CYC_SQR'], flat(g), alias=True)) == c.f12_sqr(g)
        assert unflat(run(by['CYC_SQRC'], flat(g))) == c.f12_conj(c.f12_sqr(g))
        # sparse line multiplication
        line = [rng.randrange(P) for _ in range(6)]
        sparse = ((line[0], line[1]), (0, 0), (line[2], line[3]), (line[4], line[5]), (0, 0), (0, 0))
        assert unflat(run(by['MUL_LINE'], flat(a), line, alias=True)) == c.f12_mul(a, sparse)
    return True



def psi_consts():
    """the constants of table set PT: psi's cx, cy, then the coefficients of the 11-isogeny (xnum, xden, ynum, yden; low to high)"""
    from oracle.py import iso_consts
    cx = c.f2_inv(c.f2_pow(c.XI, (P - 1) // 3))
    cy = c.f2_inv(c.f2_pow(c.XI, (P - 1) // 2))
    k = [cx[0], cx[1], cy[0], cy[1]]
    for tab in iso_consts.G1_ISO:
        k += list(tab)
    assert len(k) == TABLES.ISO_K + 12 + 11 + 16 + 16
    return k


def sim_program(ops, lay, steps, V, hook=None):
    """the device's semantics on a flat integer value store: every step reads its operand arrays at their base index;
    hook(name, k, V): the built-ins of the streamed cut (ACQ / PUB with their chunk index)"""
    by = {o.name: o for o in ops}
    cb = lay.base.get('CONST')
    if cb is not None:
        k = consts24() if 'F' in lay.base else psi_consts()        # the constants of the table set: Frobenius (F12) / psi (PT)
        V[cb:cb + len(k)] = k
    for name, d, x, y in steps:
        if name == 'FPINV':
            V[lay.ref(d)] = c.fp_inv(V[lay.ref(x)]) if V[lay.ref(x)] else 0
            continue
        if name in ('ACQ', 'PUB'):
            hook(name, d, x, V)
            continue
        if name in ('ACQF', 'PUBF'):
            hook(name, lay.ref(d), V, x) if name in getattr(hook, 'slots', ()) else hook(name, lay.ref(d), V)
            continue
        op = by[name]
        base = {DST: lay.ref(d), SA: lay.ref(x), SB: lay.ref(y), TMP: lay.base['TMP'], CONST: cb}
        at = lambda i: V[base[i >> 12] + (i & 0xfff)]  # noqa: E731
        for xa, ya, out in op.prods:
            V[base[TMP] + (out & 0xfff)] = (sum(k * at(i) for k, i in xa) % P) * (sum(k * at(i) for k, i in ya) % P) % P
        for terms, out in op.lins:
            V[base[DST] + (out & 0xfff)] = sum(k * at(i) for k, i in terms) % P


def check_programs():
    """the programs on integers against the oracle: the hard part alone, and whole pairing checks (valid and invalid)"""
    rng = random.Random(2)
    a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
    t = c.f12_mul(c.f12_conj(a), c.f12_inv(a))
    easy = c.f12_mul(c.f12_frob(t, 2), t)
    lay = layout_f12()
    B = lay.base
    V = [0] * lay.count
    V[B['F']:B['F'] + 12] = flat(easy)
    sim_program(OPS, lay, prog_final_hard(), V)
    assert unflat(V[B['T']:B['T'] + 12]) == c.final_exponentiation(a), 'hard part program'
    # whole pairing: e(P0, Q0) e(P1, Q1) with the general program, and cut into PRE_LINES / PRE_F1 / POST; the G1 points enter
    # as Jacobian triples with random Z
    sk, h = rng.randrange(1, c.R), rng.randrange(1, c.R)
    Hm = c.E1.mul(c.G1_GEN, h)
    pk = c.E2.mul(c.G2_GEN, sk)
    sig = c.E1.mul(Hm, sk)
    negg2 = c.E2.neg(c.G2_GEN)

    def jac(pt):
        z = rng.randrange(1, P)
        return [pt[0] * z * z % P, pt[1] * z * z * z % P, z, 0]

    def set_pt(V, pr, q):
        z = (rng.randrange(1, P), rng.randrange(P))
        z2 = c.f2_sqr(z)
        pt = [0] * 32
        pt[6:12] = list(c.f2_mul(q[0], z2)) + list(c.f2_mul(q[1], c.f2_mul(z2, z))) + list(z)     # Jacobian, any Z
        V[B['PT%d' % pr]:B['PT%d' % pr] + 32] = pt

    for sgn, want_one in ((sig, True), (c.E1.mul(sig, 2), False)):
        pairs = [(Hm, pk), (sgn, negg2)]
        want = c.final_exponentiation(c.miller_loop(pairs))
        V = [0] * lay.count
        V[B['F']:B['F'] + 12] = flat(c.F12_ONE)
        V[B['P']:B['P'] + 8] = jac(Hm) + jac(sgn)
        for pr, (_, q) in enumerate(pairs):
            set_pt(V, pr, q)
        sim_program(OPS, lay, dict(PROGRAMS)['PAIR_GENERAL'], V)
        assert unflat(V[B['T']:B['T'] + 12]) == want, 'pairing program'
        assert (unflat(V[B['T']:B['T'] + 12]) == c.F12_ONE) == want_one
        # the cut: three stores that share nothing but what the device hands over (pair 0's lines, the function of pair 1)
        V1 = [0] * lay.count
        set_pt(V1, 0, pk)
        V1s_pt = V1[B['PT0']:B['PT0'] + 32]
        sim_program(OPS, lay, dict(PROGRAMS)['PRE_LINES'], V1)
        V2 = [0] * lay.count
        V2[B['F']:B['F'] + 12] = flat(c.F12_ONE)
        V2[B['P'] + 4:B['P'] + 8] = jac(sgn)
        set_pt(V2, 1, negg2)
        V2g = list(V2)
        sim_program(OPS, lay, prog_key_lines(1), V2)            # stands for the table of -g2's lines
        sim_program(OPS, lay, dict(PROGRAMS)['PRE_F1'], V2)
        sim_program(OPS, lay, dict(PROGRAMS)['PRE_F1G'], V2g)
        assert V2g[B['F']:B['F'] + 12] == V2[B['F']:B['F'] + 12], 'PRE_F1G'
        V3 = [0] * lay.count
        V3[B['F']:B['F'] + 12] = flat(c.F12_ONE)
        V3[B['P']:B['P'] + 4] = V3p = jac(Hm)
        for stp in range(NSTEPS):
            o = B['L'] + 12 * stp
            V3[o:o + 6] = V1[o:o + 6]
        V3[B['W']:B['W'] + 12] = V2[B['F']:B['F'] + 12]
        sim_program(OPS, lay, dict(PROGRAMS)['POST'], V3)
        assert unflat(V3[B['T']:B['T'] + 12]) == want, 'cut pairing programs'
        # the streamed cut: PRE_LINES_S publishes chunks of eight line steps, POST_S acquires each one before its first use -- two
        # stores that share only the published chunks (the device: two workgroups of k_pairing_stream); a line read before its
        # chunk arrived would be a zero here and the result wrong
        wire, order = {}, []
        V4 = [0] * lay.count

        def pub(name, first, cnt, V):
            assert name == 'PUB' and first == sum(len(x) for x in wire.values()) and cnt > 0, 'chunks are published in order, once, without gaps'
            wire[first] = [V[B['L'] + 12 * stp:B['L'] + 12 * stp + 6] for stp in range(first, first + cnt)]
        V4[B['PT0']:B['PT0'] + 32] = V1s_pt
        sim_program(OPS, lay, dict(PROGRAMS)['PRE_LINES_S'], V4, pub)
        assert sum(len(x) for x in wire.values()) == NSTEPS and all(f % 4 == 0 and f // 4 < 32 for f in wire)      # the device's flag index
        for stp in range(NSTEPS):
            assert V4[B['L'] + 12 * stp:B['L'] + 12 * stp + 6] == V1[B['L'] + 12 * stp:B['L'] + 12 * stp + 6], 'PRE_LINES_S = PRE_LINES'
        # ... and the two consumers (the Miller loop split at SPLIT_S): each acquires exactly the chunks of its own line steps
        V5h, V5l = [0] * lay.count, [0] * lay.count
        for Vx in (V5h, V5l):
            Vx[B['F']:B['F'] + 12] = flat(c.F12_ONE)
            Vx[B['P']:B['P'] + 4] = V3p
        V5h[B['W']:B['W'] + 12] = V2[B['F']:B['F'] + 12]
        sbox = {}

        def consumer(name, a, b, V=None):
            if name == 'ACQ':
                first, cnt = a, b
                assert len(wire[first]) == cnt and first not in order, 'the consumers acquire exactly the published chunks, once'
                order.append(first)
                for j, line in enumerate(wire[first]):
                    V[B['L'] + 12 * (first + j):B['L'] + 12 * (first + j) + 6] = line
            elif name == 'PUBF':
                assert not sbox
                sbox['f'] = b[a:a + 12]
            else:
                assert name == 'ACQF'
                b[a:a + 12] = sbox['f']
        sim_program(OPS, lay, dict(PROGRAMS)['POST_LO_S'], V5l, consumer)
        lo_chunks = list(order)
        sim_program(OPS, lay, dict(PROGRAMS)['POST_HI_S'], V5h, consumer)
        assert lo_chunks == sorted(lo_chunks) and min(lo_chunks) == g.miller_line_steps(g.SPLIT_S - 1, 0)[0]
        assert sorted(order) == sorted(wire) and unflat(V5h[B['T']:B['T'] + 12]) == want, 'streamed cut programs'
        # one Miller loop on two or three workgroups: the later parts accumulate their iterations from 1, square and publish; the first
        # part takes their values from the hand-over slots and finishes the check
        for names in (('POST_HI', 'POST_LO'), ('POST3_HI', 'POST3_MID', 'POST3_LO')):
            box = {}

            def handover(name, at, V, slot):
                if name == 'PUBF':
                    assert slot not in box
                    box[slot] = V[at:at + 12]
                else:
                    V[at:at + 12] = box.pop(slot)
            handover.slots = ('ACQF', 'PUBF')
            stores = []
            for nm in names:
                Vx = [0] * lay.count
                Vx[B['F']:B['F'] + 12] = flat(c.F12_ONE)
                Vx[B['P']:B['P'] + 4] = V3p
                for stp in range(NSTEPS):
                    o = B['L'] + 12 * stp
                    Vx[o:o + 6] = V1[o:o + 6]
                stores.append(Vx)
            stores[0][B['W']:B['W'] + 12] = V2[B['F']:B['F'] + 12]
            for nm, Vx in list(zip(names, stores))[:0:-1]:
                sim_program(OPS, lay, dict(PROGRAMS)[nm], Vx, handover)
            assert sorted(box) == list(range(len(names) - 1)), 'every later part publishes into its own slot'
            sim_program(OPS, lay, dict(PROGRAMS)[names[0]], stores[0], handover)
            assert not box and unflat(stores[0][B['T']:B['T'] + 12]) == want, 'split Miller loop programs %s' % names[0]
    # the fold tree's sixteen-way product
    vals = [tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6)) for _ in range(16)]
    Vt = [0] * lay.count
    for j, x in enumerate(vals):
        Vt[B['L'] + 12 * j:B['L'] + 12 * j + 12] = flat(x)
    sim_program(OPS, lay, dict(PROGRAMS)['F12_TREE16'], Vt)
    want = vals[0]
    for x in vals[1:]:
        want = c.f12_mul(want, x)
    assert unflat(Vt[B['L']:B['L'] + 12]) == want, 'F12_TREE16'
    # the Horner chain of a pairing product's per-entry line products: entries 1, 4, 8, 18, 51 are the addition steps (no squaring)
    vals = [tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6)) for _ in range(NSTEPS)]
    Vh = [0] * lay.count
    for j, x in enumerate(vals):
        Vh[B['L'] + 12 * j:B['L'] + 12 * j + 12] = flat(x)
    sim_program(OPS, lay, dict(PROGRAMS)['HORNER'], Vh)
    want = vals[0]
    for e in range(1, NSTEPS):
        if e not in (1, 4, 8, 18, 51):
            want = c.f12_mul(want, want)
        want = c.f12_mul(want, vals[e])
    assert unflat(Vh[B['F']:B['F'] + 12]) == c.f12_conj(want), 'HORNER'
    return lay


def check_point_programs():
    """the point-sum programs against the oracle's affine additions: random points, repeated points, opposite points, identities,
    random projective scalings of the inputs, every input / output coordinate combination"""
    rng = random.Random(3)
    lay = layout_pt()
    for g in (1, 2):
        E, gen, S = (c.E1, c.G1_GEN, G1S) if g == 1 else (c.E2, c.G2_GEN, G2S)
        w = 1 if g == 1 else 2                      # values per coordinate
        fmul = (lambda x, y: x * y % P) if g == 1 else c.f2_mul
        vals = (lambda x: [x]) if g == 1 else (lambda x: list(x))
        one, zero = (1, 0) if g == 1 else (c.F2_ONE, c.F2_ZERO)
        for case in range(4):
            pts = [E.mul(gen, rng.randrange(1, c.R)) for _ in range(PT_POINTS)]
            if case == 1:
                pts[1] = pts[0]                     # a doubling
                pts[3] = E.neg(pts[2])              # a cancellation
                pts[5] = None                       # identities on either side and against each other
                pts[6] = None
                pts[7] = None
                pts[8:12] = [pts[8]] * 4
            if case == 2:
                pts = [None] * PT_POINTS
            if case == 3:
                pts = [pts[0], E.neg(pts[0])] * (PT_POINTS // 2)
            want = None
            for q in pts:
                want = E.add(want, q)
            for jin in (1, 0):
                for jout in (1, 0):
                    V = [0] * lay.count
                    for i, q in enumerate(pts):
                        lam = rng.randrange(1, P) if g == 1 else (rng.randrange(1, P), rng.randrange(P))
                        if q is None:
                            X, Y, Z = (lam, one, zero) if jin else (zero, lam, zero)
                            if jin:
                                X = zero           # the loader's normal form of a Jacobian identity: (0, 1, 0)
                        elif jin:                   # Jacobian (x l^2, y l^3, l)
                            l2 = fmul(lam, lam)
                            X, Y, Z = fmul(q[0], l2), fmul(q[1], fmul(l2, lam)), lam
                        else:                       # homogeneous (x l, y l, l)
                            X, Y, Z = fmul(q[0], lam), fmul(q[1], lam), lam
                        o = lay.base['L0'] + (g2_point_off(i) if g == 2 else g1_point_off(i))
                        V[o:o + 3 * w] = vals(X) + vals(Y) + vals(Z)
                    name = 'G%d_%s%s' % (g, 'J' if jin else 'H', 'J' if jout else 'H')
                    sim_program(OPS_PT, lay, dict(PROGRAMS_PT)[name], V)
                    o = lay.base['L4']
                    if g == 1:
                        X, Y, Z = V[o], V[o + 1], V[o + 2]
                        inv, is0 = c.fp_inv, (lambda z: z == 0)
                    else:
                        X, Y, Z = (V[o], V[o + 1]), (V[o + 2], V[o + 3]), (V[o + 4], V[o + 5])
                        inv, is0 = c.f2_inv, (lambda z: z == (0, 0))
                    if is0(Z):
                        got = None
                    elif jout:
                        zi = inv(Z)
                        zi2 = fmul(zi, zi)
                        got = (fmul(X, zi2), fmul(Y, fmul(zi2, zi)))
                    else:
                        zi = inv(Z)
                        got = (fmul(X, zi), fmul(Y, zi))
                    assert got == want, (name, case)
    # the cofactor clearing of hash-to-G2: points of E2(Fp2) OUTSIDE the subgroup (sums of two mapped points), the identity, a
    # point already in G2; Jacobian in and out
    B = lay.base
    for case in range(4):
        if case < 2:
            q = c.E2.add(c.map_to_curve_g2((rng.randrange(P), rng.randrange(P))), c.map_to_curve_g2((rng.randrange(P), rng.randrange(P))))
        elif case == 2:
            q = c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
        else:
            q = None
        want = c.g2_clear_cofactor(q) if q is not None else None
        V = [0] * lay.count
        if q is None:
            V[B['R0']:B['R0'] + 6] = [0, 0, 1, 0, 0, 0]
        else:
            z = (rng.randrange(1, P), rng.randrange(P))
            z2 = c.f2_sqr(z)
            V[B['R0']:B['R0'] + 6] = list(c.f2_mul(q[0], z2)) + list(c.f2_mul(q[1], c.f2_mul(z2, z))) + list(z)
        sim_program(OPS_PT, lay, dict(PROGRAMS_PT)['G2_CLEAR'], V)
        o = B['R3']
        X, Y, Z = (V[o], V[o + 1]), (V[o + 2], V[o + 3]), (V[o + 4], V[o + 5])
        if Z == (0, 0):
            got = None
        else:
            zi = c.f2_inv(Z)
            zi2 = c.f2_sqr(zi)
            got = (c.f2_mul(X, zi2), c.f2_mul(Y, c.f2_mul(zi2, zi)))
        assert got == want, ('G2_CLEAR', case)
    # the tail of hash-to-G2: two mapped points (homogeneous, any scaling; the second one may be the identity) -> h_eff (q0 + q1)
    for case in range(3):
        q0 = c.map_to_curve_g2((rng.randrange(P), rng.randrange(P)))
        q1 = c.map_to_curve_g2((rng.randrange(P), rng.randrange(P))) if case < 2 else None
        want = c.g2_clear_cofactor(c.E2.add(q0, q1))
        V = [0] * lay.count
        for reg, q in (('R0', q0), ('R1', q1)):
            if q is None:
                V[B[reg]:B[reg] + 6] = [0, 0, 1, 0, 0, 0]
            else:
                z = (rng.randrange(1, P), rng.randrange(P))
                V[B[reg]:B[reg] + 6] = list(c.f2_mul(q[0], z)) + list(c.f2_mul(q[1], z)) + list(z)
        sim_program(OPS_PT, lay, dict(PROGRAMS_PT)['G2_HASH_TAIL'], V)
        o = B['R3']
        X, Y, Z = (V[o], V[o + 1]), (V[o + 2], V[o + 3]), (V[o + 4], V[o + 5])
        zi = c.f2_inv(Z)
        zi2 = c.f2_sqr(zi)
        assert (c.f2_mul(X, zi2), c.f2_mul(Y, c.f2_mul(zi2, zi))) == want, ('G2_HASH_TAIL', case)
    # the cofactor clearing of hash-to-G1: (1 - x) P for points of E1(Fp) outside the subgroup, in it, and the identity
    for case in range(4):
        if case < 2:
            q = c.E1.add(c.map_to_curve_g1(rng.randrange(P)), c.map_to_curve_g1(rng.randrange(P)))
        elif case == 2:
            q = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R))
        else:
            q = None
        want = c.E1.mul(q, c.H_EFF_G1) if q is not None else None
        V = [0] * lay.count
        if q is None:
            V[B['R0']:B['R0'] + 3] = [0, 1, 0]
        else:
            z = rng.randrange(1, P)
            V[B['R0']:B['R0'] + 3] = [q[0] * z * z % P, q[1] * z * z * z % P, z]
        sim_program(OPS_PT, lay, dict(PROGRAMS_PT)['G1_CLEAR'], V)
        X, Y, Z = V[B['R1']:B['R1'] + 3]
        if Z == 0:
            got = None
        else:
            zi = c.fp_inv(Z)
            got = (X * zi * zi % P, Y * zi * zi * zi % P)
        assert got == want, ('G1_CLEAR', case)
    # everything behind the two SSWU maps of hash-to-G1 (isogeny of both points, sum, clearing) against the oracle's own chain;
    # x' enters as a fraction xn / xd with a random denominator
    from oracle.py import iso_consts
    for case in range(3):
        us = [rng.randrange(P), rng.randrange(P)]
        qs = [c._sswu(c.E1_ISO, iso_consts.G1_Z, u, c.fp_is_square, c.fp_sqrt, c._sgn0_fp, 1) for u in us]
        want = c.E1.mul(c.E1.add(c.map_to_curve_g1(us[0]), c.map_to_curve_g1(us[1])), c.H_EFF_G1)
        V = [0] * lay.count
        for m, q in enumerate(qs):
            d = rng.randrange(1, P)
            o = B['ISO'] + TABLES.ISO_STRIDE * m
            V[o + TABLES.ISO_XN], V[o + TABLES.ISO_XD], V[o + TABLES.ISO_Y] = q[0] * d % P, d, q[1]
        Vm = list(V)
        sim_program(OPS_PT, lay, dict(PROGRAMS_PT)['G1_HASH_TAIL'], V)
        X, Y, Z = V[B['R1']:B['R1'] + 3]
        zi = c.fp_inv(Z)
        assert (X * zi * zi % P, Y * zi * zi * zi % P) == want, ('G1_HASH_TAIL', case)
        sim_program(OPS_PT, lay, dict(PROGRAMS_PT)['G1_HASH_MAP'], Vm)          # the same point before the clearing, in R0
        X, Y, Z = Vm[B['R0']:B['R0'] + 3]
        zi = c.fp_inv(Z)
        assert c.E1.mul((X * zi * zi % P, Y * zi * zi * zi % P), c.H_EFF_G1) == want, ('G1_HASH_MAP', case)
    return lay
