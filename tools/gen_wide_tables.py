"""Generates agora-blsful_amd/csrc/wide_tables.cuh: the operation tables of the row-wide engine (csrc/wide_engine.cuh).

An operation is a list of PRODUCT sub-rounds -- up to 16 rows each computing  out = (sum ca_i V[a_i]) * (sum cb_i V[b_i])  in
Fp with at most two terms per operand -- followed by one LINEAR phase in which up to 16 rows each write
V[out] = reduce(sum c_i V[idx_i]).  Symbolic value indices are (slot << 12 | offset) with the slots 0 = destination,
1 = first operand, 2 = second operand, 3 = product scratch, 4 = constants; a PROGRAM is a list of (operation, destination
array, operand arrays) steps, e.g. the hard part of the final exponentiation.

The tables are derived here from the tower formulas (Fp12 over the basis w^0..w^5, w^6 = xi = 1 + u, every coefficient an
Fp2 = (re, im)) and CHECKED before they are written: a plain-integer simulation of the engine's semantics runs every
operation on random inputs and compares with the oracle's Fp12 arithmetic (oracle/py/bls381.py).  Run:
    python tools/gen_wide_tables.py            (writes the header; exits non-zero if a check fails)
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.py import bls381 as c  # noqa: E402

P = c.P
DST, SA, SB, TMP, CONST = 0, 1, 2, 3, 4
MAXLIN = 18


def idx(slot, off):
    return (slot << 12) | off


class Op:
    def __init__(self, name):
        self.name = name
        self.prods = []      # (a terms [(coef, idx)], b terms, out idx)
        self.lins = []       # (terms [(coef, idx)], out idx)
        self.ntmp = 0

    def tmp(self):
        self.ntmp += 1
        return idx(TMP, self.ntmp - 1)

    def prod(self, a, b):
        assert 1 <= len(a) <= 2 and 1 <= len(b) <= 2
        t = self.tmp()
        self.prods.append((a, b, t))
        return t

    def lin(self, terms, out):
        terms = [(k, i) for k, i in terms if k]
        assert len(terms) <= MAXLIN, (self.name, len(terms))
        self.lins.append((terms, out))

    # Fp2 helpers on symbolic values: an Fp2 is a pair of term lists (re, im), each a list of (coef, idx)
    def fp2_mul(self, x, y):
        """Karatsuba: three Fp products; x, y = ((coef, idx) re, (coef, idx) im) single-term components"""
        (xr, xi), (yr, yi) = x, y
        p0 = self.prod([xr], [yr])
        p1 = self.prod([xi], [yi])
        p2 = self.prod([xr, xi], [yr, yi])
        return ([(1, p0), (-1, p1)], [(1, p2), (-1, p0), (-1, p1)])

    def fp2_sqr(self, x):
        (xr, xi) = x
        s1 = self.prod([xr, xi], [xr, (-xi[0], xi[1])])
        s2 = self.prod([xr], [xi])
        return ([(1, s1)], [(2, s2)])


def scale(terms, k):
    return [(k * a, i) for a, i in terms]


def mul_xi(re, im):
    """(re + im u)(1 + u) = (re - im) + (re + im) u on term lists"""
    return (re + scale(im, -1), re + im)


def merge(terms):
    acc = {}
    for k, i in terms:
        acc[i] = acc.get(i, 0) + k
    return [(k, i) for i, k in acc.items() if k]


def coef(slot, k):
    return ((1, idx(slot, 2 * k)), (1, idx(slot, 2 * k + 1)))


def op_mul():
    op = Op('MUL')                       # dst = a * b
    out = [([], []) for _ in range(6)]
    for i in range(6):
        for j in range(6):
            re, im = op.fp2_mul(coef(SA, i), coef(SB, j))
            k = i + j
            if k >= 6:
                re, im = mul_xi(re, im)
                k -= 6
            out[k] = (out[k][0] + re, out[k][1] + im)
    for k in range(6):
        op.lin(merge(out[k][0]), idx(DST, 2 * k))
        op.lin(merge(out[k][1]), idx(DST, 2 * k + 1))
    return op


def op_sqr():
    op = Op('SQR')                       # dst = a^2 (general)
    out = [([], []) for _ in range(6)]
    for i in range(6):
        for j in range(i, 6):
            if i == j:
                re, im = op.fp2_sqr(coef(SA, i))
            else:
                re, im = op.fp2_mul(coef(SA, i), coef(SA, j))
                re, im = scale(re, 2), scale(im, 2)
            k = i + j
            if k >= 6:
                re, im = mul_xi(re, im)
                k -= 6
            out[k] = (out[k][0] + re, out[k][1] + im)
    for k in range(6):
        op.lin(merge(out[k][0]), idx(DST, 2 * k))
        op.lin(merge(out[k][1]), idx(DST, 2 * k + 1))
    return op


def op_cyc_sqr():
    """Granger-Scott squaring in the cyclotomic subgroup; the Fp4 pairs are (c0, c3), (c1, c4), (c2, c5):
    (a + b s)^2 = (a^2 + xi b^2) + 2ab s;  new = 3 t - 2 z (even coefficients) / 3 t + 2 z (odd)"""
    op = Op('CYC_SQR')
    halves = {}
    for m in range(3):
        a, b = coef(SA, m), coef(SA, m + 3)
        A = op.fp2_sqr(a)
        B = op.fp2_sqr(b)
        AB = op.fp2_mul(a, b)
        xb = mul_xi(*B)
        even = (A[0] + xb[0], A[1] + xb[1])
        odd = (scale(AB[0], 2), scale(AB[1], 2))
        halves[m] = (even, odd)
    target = {0: (0, 0), 3: (0, 1), 2: (1, 0), 5: (1, 1), 4: (2, 0), 1: (2, 1)}
    for k in range(6):
        m, h = target[k]
        t = halves[m][h]
        if k == 1:
            t = mul_xi(*t)
        sgn = 2 if (k & 1) else -2
        for comp in range(2):
            z = idx(SA, 2 * k + comp)
            op.lin(merge(scale(t[comp], 3) + [(sgn, z)]), idx(DST, 2 * k + comp))
    return op


def op_frob(j):
    """dst = a^(p^j): coefficient k -> conj^j(c_k) * FROBj[k]; constants at CONST + 12 (j - 1) + 2k (+1)"""
    op = Op('FROB%d' % j)
    for k in range(6):
        re, im = coef(SA, k)
        if j == 1:
            im = (-1, im[1])
        if k == 0:
            op.lin([re], idx(DST, 0))
            op.lin([im], idx(DST, 1))
            continue
        kc = ((1, idx(CONST, 12 * (j - 1) + 2 * k)), (1, idx(CONST, 12 * (j - 1) + 2 * k + 1)))
        r, i = op.fp2_mul((re, im), kc)
        op.lin(merge(r), idx(DST, 2 * k))
        op.lin(merge(i), idx(DST, 2 * k + 1))
    return op


def op_conj():
    op = Op('CONJ')                      # a^(p^6): negate the odd coefficients
    for k in range(6):
        for comp in range(2):
            op.lin([(-1 if k & 1 else 1, idx(SA, 2 * k + comp))], idx(DST, 2 * k + comp))
    return op


def op_copy():
    op = Op('COPY')
    for k in range(12):
        op.lin([(1, idx(SA, k))], idx(DST, k))
    return op


OPS = [op_mul(), op_sqr(), op_cyc_sqr(), op_frob(1), op_frob(2), op_conj(), op_copy()]


# ------------------------------------------------------------------ simulation of the engine on integers
def run(op, a12, b12=None, alias=False):
    consts = []
    for j in (1, 2):
        for k in range(6):
            consts += list(c.f2_pow(c.XI, k * (P ** j - 1) // 6))
    slots = {SA: list(a12), SB: list(b12 or [0] * 12), CONST: consts, DST: [0] * 12}
    if alias:
        slots[DST] = slots[SA]
    # the device lets every linear row read its inputs and write its output without a barrier in between: with dst aliasing
    # a source that is only safe when no row reads a value another row writes -- checked here by writing row by row
    V = lambda i: slots[i >> 12][i & 0xfff]  # noqa: E731
    slots[TMP] = [0] * op.ntmp
    for a, b, out in op.prods:
        x = sum(k * V(i) for k, i in a) % P
        y = sum(k * V(i) for k, i in b) % P
        slots[TMP][out & 0xfff] = x * y % P
    for terms, out in op.lins:
        slots[out >> 12][out & 0xfff] = sum(k * V(i) for k, i in terms) % P
    return slots[DST]


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
    return True


# ------------------------------------------------------------------ programs: sequences of (op, dst, a, b) over named value arrays
ARR = {'F': 0, 'T': 12, 'U': 24, 'W': 36, 'ACC': 48}
X_ABS = 0xd201000000010000


def prog_pow_x(dst, a):
    """dst = a^x (x < 0) in the cyclotomic subgroup"""
    st = [('COPY', 'ACC', a, a)]
    for i in range(62, -1, -1):
        st.append(('CYC_SQR', 'ACC', 'ACC', 'ACC'))
        if (X_ABS >> i) & 1:
            st.append(('MUL', 'ACC', 'ACC', a))
    st.append(('CONJ', dst, 'ACC', 'ACC'))
    return st


def prog_final_hard():
    """T <- F^((x-1)^2 (x+p) (x^2+p^2-1)) * F^3 (the chain of pairing.cuh's final_exponentiation)"""
    st = prog_pow_x('T', 'F')
    st += [('CONJ', 'U', 'F', 'F'), ('MUL', 'T', 'T', 'U')]            # f^(x-1)
    st += prog_pow_x('U', 'T')
    st += [('CONJ', 'W', 'T', 'T'), ('MUL', 'T', 'U', 'W')]            # f^((x-1)^2)
    st += prog_pow_x('U', 'T')
    st += [('FROB1', 'W', 'T', 'T'), ('MUL', 'T', 'U', 'W')]           # ^(x+p)
    st += prog_pow_x('U', 'T')
    st += prog_pow_x('U', 'U')                                          # t^(x^2)
    st += [('FROB2', 'W', 'T', 'T'), ('MUL', 'U', 'U', 'W'), ('CONJ', 'W', 'T', 'T'), ('MUL', 'T', 'U', 'W')]   # ^(x^2+p^2-1)
    st += [('SQR', 'U', 'F', 'F'), ('MUL', 'U', 'U', 'F'), ('MUL', 'T', 'T', 'U')]                              # * f^3
    return st


PROGRAMS = [('FINAL_HARD', prog_final_hard())]


def check_programs():
    """the hard-part program on integers against the oracle's final exponentiation"""
    rng = random.Random(2)
    by = {o.name: o for o in OPS}
    a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
    t = c.f12_mul(c.f12_conj(a), c.f12_inv(a))
    easy = c.f12_mul(c.f12_frob(t, 2), t)
    store = {k: [0] * 12 for k in ARR}
    store['F'] = flat(easy)
    for name, d, x, y in prog_final_hard():
        res = run(by[name], list(store[x]), list(store[y]))
        store[d] = list(res)
    assert unflat(store['T']) == c.final_exponentiation(a), 'hard part program'
    return True


# ------------------------------------------------------------------ emission
def emit(path):
    out = []
    out.append('// GENERATED by tools/gen_wide_tables.py -- do not edit.  Operation tables and programs of the row-wide engine')
    out.append('// (wide_engine.cuh).  Every table and program was checked against the oracle\'s Fp12 arithmetic by the generator.')
    out.append('#pragma once')
    out.append('// product row: out = (ca0 V[A + a0] + ca1 V[A + a1]) * (cb0 V[B + b0] + cb1 V[B + b1]) -> V[TMP + out]; A, B: the op\'s operand arrays')
    out.append('struct wide_prod { uint8_t a[2], b[2]; int8_t ca[2], cb[2]; uint8_t out, pad[3]; };')
    out.append('// linear row: V[DST + out] = reduce(sum c_i V[idx_i]); idx bit 7 set: operand array A, clear: product scratch')
    out.append('struct wide_lin { uint8_t n, out; uint8_t idx[%d]; int8_t c[%d]; uint8_t pad[2]; };' % (MAXLIN, MAXLIN))
    out.append('struct wide_op { uint16_t prod_off, lin_off; uint8_t nsub, nlin, b_is_const, b_is_a; };')
    prods, lins, ops = [], [], []
    for op in OPS:
        nsub = (len(op.prods) + 15) // 16
        assert len(op.lins) <= 16 and op.ntmp < 127
        bslots = {i >> 12 for _, b, _ in op.prods for _, i in b}
        aslots = {i >> 12 for a, _, _ in op.prods for _, i in a}
        assert aslots <= {SA} and len(bslots) <= 1 and bslots <= {SA, SB, CONST}, (op.name, aslots, bslots)
        ops.append((op.name, len(prods), len(lins), nsub, len(op.lins), 1 if bslots == {CONST} else 0, 1 if bslots == {SA} else 0, len(op.prods)))
        rows = list(op.prods) + [None] * (nsub * 16 - len(op.prods))
        for r in rows:
            if r is None:      # an idle row multiplies zero by zero into a scratch value that nothing reads
                prods.append('{{0, 0}, {0, 0}, {0, 0}, {0, 0}, %d, {0, 0, 0}}' % op.ntmp)
                continue
            a, b, o = r
            assert (o >> 12) == TMP
            a = a + [(0, 0)] * (2 - len(a))
            b = b + [(0, 0)] * (2 - len(b))
            prods.append('{{%d, %d}, {%d, %d}, {%d, %d}, {%d, %d}, %d, {0, 0, 0}}' % (a[0][1] & 0xfff, a[1][1] & 0xfff, b[0][1] & 0xfff, b[1][1] & 0xfff,
                                                                                   a[0][0], a[1][0], b[0][0], b[1][0], o & 0xfff))
        for terms, o in op.lins:
            assert (o >> 12) == DST
            ii, cc = [], []
            for k, i in terms:
                assert (i >> 12) in (TMP, SA) and (i & 0xfff) < 128 and -128 < k < 128
                ii.append((0x80 if (i >> 12) == SA else 0) | (i & 0xfff))
                cc.append(k)
            ii += [0] * (MAXLIN - len(ii))
            cc += [0] * (MAXLIN - len(cc))
            lins.append('{%d, %d, {%s}, {%s}, {0, 0}}' % (len(terms), o & 0xfff, ', '.join(map(str, ii)), ', '.join(map(str, cc))))
    out.append('BLS_CONST wide_prod WIDE_PROD[%d] = {' % len(prods))
    out += ['    %s,' % p for p in prods]
    out.append('};')
    out.append('BLS_CONST wide_lin WIDE_LIN[%d] = {' % len(lins))
    out += ['    %s,' % l for l in lins]
    out.append('};')
    out.append('BLS_CONST wide_op WIDE_OPS[%d] = {' % len(ops))
    for name, po, lo, ns, nl, bc, ba, np_ in ops:
        out.append('    {%d, %d, %d, %d, %d, %d},   // WOP_%s: %d products' % (po, lo, ns, nl, bc, ba, name, np_))
    out.append('};')
    names = [o[0] for o in ops]
    for k, name in enumerate(names):
        out.append('#define WOP_%s %d' % (name, k))
    out.append('#define WIDE_MAX_TMP %d' % (max(o.ntmp for o in OPS) + 1))
    for k, v in ARR.items():
        out.append('#define WV_%s %d' % (k, v))
    out.append('// programs: op | dst << 8 | a << 16 | b << 24 (value-array bases)')
    for pname, st in PROGRAMS:
        words = ['0x%08xu' % (names.index(n) | ARR[d] << 8 | ARR[x] << 16 | ARR[y] << 24) for n, d, x, y in st]
        out.append('#define WIDE_PROG_%s_LEN %d' % (pname, len(words)))
        out.append('BLS_CONST uint32_t WIDE_PROG_%s[%d] = {' % (pname, len(words)))
        for i in range(0, len(words), 8):
            out.append('    ' + ', '.join(words[i:i + 8]) + ',')
        out.append('};')
    open(path, 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
    assert self_check() and check_programs()
    emit(os.path.join(ROOT, 'agora-blsful_amd', 'csrc', 'wide_tables.cuh'))
    for o in OPS:
        print(o.name, len(o.prods), 'products,', len(o.lins), 'linear rows, longest', max((len(t) for t, _ in o.lins), default=0))
