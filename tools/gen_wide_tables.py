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
ROWS = int(os.environ.get('WIDE_ROWS', '16'))     # DPP rows of the workgroup = products per sub-round (16: four waves, one per SIMD)


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


# ------------------------------------------------------------------ Miller loop on the engine
# Point workspace of one pair (array PT, 32 values): T = X Y Z (0..5), Q = xq yq (6..9), line l0 c2 c3 (10..15), M (16..31).
# The steps are the homogeneous-projective formulas of csrc/pairing.cuh (miller_dbl_step / miller_add_step), cut into levels
# of independent products; the line leaves UNSCALED (c2, c3 still to be multiplied by xP, yP: LSCALE).
PT_T, PT_Q, PT_L, PT_M = 0, 6, 10, 16


def f2(slot, off):
    return ((1, idx(slot, off)), (1, idx(slot, off + 1)))


def lin2(op, val, out_off, k=1):
    """write the Fp2 term-list pair `val` scaled by k to dst[out_off], dst[out_off + 1]"""
    op.lin(merge(scale(val[0], k)), idx(DST, out_off))
    op.lin(merge(scale(val[1], k)), idx(DST, out_off + 1))


def sub2(x, y):
    return (x[0] + scale(y[0], -1), x[1] + scale(y[1], -1))


def add2(x, y):
    return (x[0] + y[0], x[1] + y[1])


def op_pdbl1():
    op = Op('PDBL1')
    X, Y, Z = f2(SA, PT_T), f2(SA, PT_T + 2), f2(SA, PT_T + 4)
    A = op.fp2_mul(X, Y)
    B = op.fp2_sqr(Y)
    C = op.fp2_sqr(Z)
    Hh = op.fp2_mul(Y, Z)
    S = op.fp2_sqr(X)
    xc = mul_xi(*C)
    E = (scale(xc[0], 12), scale(xc[1], 12))            # 3b' C = 12 (1 + u) C
    F = (scale(E[0], 3), scale(E[1], 3))
    H = (scale(Hh[0], 2), scale(Hh[1], 2))
    lin2(op, A, PT_M + 0)
    lin2(op, sub2(B, F), PT_M + 2)
    lin2(op, add2(B, F), PT_M + 4)
    lin2(op, E, PT_M + 6)
    lin2(op, B, PT_M + 8)
    lin2(op, H, PT_M + 10)
    lin2(op, sub2(B, E), PT_L + 0)                        # l0 = Y^2 - 3b'Z^2
    lin2(op, S, PT_L + 2, -3)                             # c2 = -3 X^2
    lin2(op, H, PT_L + 4)                                 # c3 = 2YZ
    return op


def op_pdbl2():
    op = Op('PDBL2')
    M = [f2(SA, PT_M + 2 * k) for k in range(6)]
    X3 = op.fp2_mul(M[0], M[1])                           # A (B - F)
    Y3 = op.fp2_sqr(M[2])                                 # (B + F)^2
    EE = op.fp2_sqr(M[3])
    Z3 = op.fp2_mul(M[4], M[5])                           # B H
    lin2(op, X3, PT_T + 0, 2)
    lin2(op, sub2(Y3, (scale(EE[0], 12), scale(EE[1], 12))), PT_T + 2)
    lin2(op, Z3, PT_T + 4, 4)
    return op


def op_padd1():
    op = Op('PADD1')
    X, Y, Z = f2(SA, PT_T), f2(SA, PT_T + 2), f2(SA, PT_T + 4)
    xq, yq = f2(SA, PT_Q), f2(SA, PT_Q + 2)
    yz = op.fp2_mul(yq, Z)
    xz = op.fp2_mul(xq, Z)
    lin2(op, sub2(([Y[0]], [Y[1]]), yz), PT_M + 0)        # theta = Y - yq Z
    lin2(op, sub2(([X[0]], [X[1]]), xz), PT_M + 2)        # lambda = X - xq Z
    return op


def op_padd2():
    op = Op('PADD2')
    th, la = f2(SA, PT_M + 0), f2(SA, PT_M + 2)
    xq, yq = f2(SA, PT_Q), f2(SA, PT_Q + 2)
    a = op.fp2_mul(th, xq)
    b = op.fp2_mul(la, yq)
    cc = op.fp2_sqr(th)
    dd = op.fp2_sqr(la)
    lin2(op, sub2(a, b), PT_L + 0)                        # l0 = theta xq - lambda yq
    lin2(op, ([th[0]], [th[1]]), PT_L + 2, -1)            # c2 = -theta
    lin2(op, ([la[0]], [la[1]]), PT_L + 4)                # c3 = lambda
    lin2(op, cc, PT_M + 4)
    lin2(op, dd, PT_M + 6)
    return op


def op_padd3():
    op = Op('PADD3')
    la, cc, dd = f2(SA, PT_M + 2), f2(SA, PT_M + 4), f2(SA, PT_M + 6)
    X, Z = f2(SA, PT_T), f2(SA, PT_T + 4)
    e = op.fp2_mul(la, dd)
    f = op.fp2_mul(Z, cc)
    g = op.fp2_mul(X, dd)
    h = sub2(add2(e, f), (scale(g[0], 2), scale(g[1], 2)))
    lin2(op, e, PT_M + 8)
    lin2(op, sub2(g, h), PT_M + 10)                       # g - h
    lin2(op, h, PT_M + 12)
    return op


def op_padd4():
    op = Op('PADD4')
    th, la = f2(SA, PT_M + 0), f2(SA, PT_M + 2)
    e, gh, h = f2(SA, PT_M + 8), f2(SA, PT_M + 10), f2(SA, PT_M + 12)
    Y, Z = f2(SA, PT_T + 2), f2(SA, PT_T + 4)
    x3 = op.fp2_mul(la, h)
    t = op.fp2_mul(th, gh)
    u = op.fp2_mul(e, Y)
    z3 = op.fp2_mul(Z, e)
    lin2(op, x3, PT_T + 0)
    lin2(op, sub2(t, u), PT_T + 2)
    lin2(op, z3, PT_T + 4)
    return op


def op_copy6():
    op = Op('COPY6')                                      # dst[0..5] = the line of the point workspace
    for k in range(6):
        op.lin([(1, idx(SA, PT_L + k))], idx(DST, k))
    return op


def op_lscale():
    """the lines of both pairs of one step (array L[step]: pair p at 6p) scaled by (xP, yP) of their pair (array P: 2p, 2p+1)"""
    op = Op('LSCALE')
    for pr in range(2):
        for part, pc in ((2, 0), (4, 1)):                 # c2 * xP, c3 * yP
            for comp in range(2):
                t = op.prod([(1, idx(SA, 6 * pr + part + comp))], [(1, idx(SB, 2 * pr + pc))])
                op.lin([(1, t)], idx(DST, 6 * pr + part + comp))
    return op


def op_mul_line():
    """dst = a * (l0 + l2 w^2 + l3 w^3), the line's three Fp2 at b[0..5]"""
    op = Op('MUL_LINE')
    out = [([], []) for _ in range(6)]
    for i in range(6):
        for jw, lo in ((0, 0), (2, 2), (3, 4)):
            re, im = op.fp2_mul(coef(SA, i), f2(SB, lo))
            k = i + jw
            if k >= 6:
                re, im = mul_xi(re, im)
                k -= 6
            out[k] = (out[k][0] + re, out[k][1] + im)
    for k in range(6):
        op.lin(merge(out[k][0]), idx(DST, 2 * k))
        op.lin(merge(out[k][1]), idx(DST, 2 * k + 1))
    return op


OPS += [op_pdbl1(), op_pdbl2(), op_padd1(), op_padd2(), op_padd3(), op_padd4(), op_copy6(), op_lscale(), op_mul_line()]


# ------------------------------------------------------------------ simulation of the engine on integers
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
        # sparse line multiplication
        line = [rng.randrange(P) for _ in range(6)]
        sparse = ((line[0], line[1]), (0, 0), (line[2], line[3]), (line[4], line[5]), (0, 0), (0, 0))
        assert unflat(run(by['MUL_LINE'], flat(a), line, alias=True)) == c.f12_mul(a, sparse)
    return True


# ------------------------------------------------------------------ programs: sequences of (op, dst, a, b) over the value store
X_ABS = 0xd201000000010000
NSTEPS = 68                      # Miller steps: 63 doublings interleaved with 5 additions


class Layout:
    """value-store indices (16-word values)"""
    def __init__(self, ntmp):
        self.base = {}
        off = 0
        for name, n in (('F', 12), ('T', 12), ('U', 12), ('W', 12), ('ACC', 12), ('TMP', ntmp), ('CONST', 24), ('P', 4), ('PT0', 32), ('PT1', 32),
                        ('L', 12 * NSTEPS)):
            self.base[name] = off
            off += n
        self.count = off

    def ref(self, r):
        """'F' or ('L', 12) -> value index"""
        if isinstance(r, tuple):
            return self.base[r[0]] + r[1]
        return self.base[r]


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


def prog_key_lines(pair):
    """the UNSCALED line coefficients of the Miller steps of pair `pair` (its point workspace holds T = Q, Z = 1) -> L[step][pair]"""
    pt = 'PT%d' % pair
    st, step = [], 0
    for i in range(62, -1, -1):
        st += [('PDBL1', pt, pt, pt), ('PDBL2', pt, pt, pt), ('COPY6', ('L', 12 * step + 6 * pair), pt, pt)]
        step += 1
        if (X_ABS >> i) & 1:
            st += [('PADD%d' % k, pt, pt, pt) for k in (1, 2, 3, 4)] + [('COPY6', ('L', 12 * step + 6 * pair), pt, pt)]
            step += 1
    assert step == NSTEPS
    return st


def prog_miller():
    """F <- conj(prod of the two pairs' Miller functions), the lines taken from L (scaled here by the pairs' G1 points)"""
    st, step = [], 0
    for i in range(62, -1, -1):
        if i != 62:
            st.append(('SQR', 'F', 'F', 'F'))
        for _ in range(2 if (X_ABS >> i) & 1 else 1):
            ls = ('L', 12 * step)
            st += [('LSCALE', ls, ls, 'P'), ('MUL_LINE', 'F', 'F', ls), ('MUL_LINE', 'F', 'F', ('L', 12 * step + 6))]
            step += 1
    st.append(('CONJ', 'F', 'F', 'F'))
    return st


def prog_easy():
    """F <- F^((p^6 - 1)(p^2 + 1)); INV is the interpreter's lane-local Fp12 inversion T <- F^-1"""
    return [('CONJ', 'U', 'F', 'F'), ('INV', 'T', 'F', 'F'), ('MUL', 'F', 'U', 'T'), ('FROB2', 'T', 'F', 'F'), ('MUL', 'F', 'T', 'F')]


PROGRAMS = [('FINAL_HARD', prog_final_hard()),
            # core_verify of Bls12381G1Impl: pair 1's G2 argument is the constant -g2, its lines come from a table
            ('PAIR_FIXED', prog_key_lines(0) + prog_miller() + prog_easy() + prog_final_hard()),
            ('PAIR_GENERAL', prog_key_lines(0) + prog_key_lines(1) + prog_miller() + prog_easy() + prog_final_hard())]


def sim_program(steps, store):
    by = {o.name: o for o in OPS}

    def arr(r):
        if isinstance(r, tuple):
            return store[r[0]], r[1]
        return store[r], 0

    class View(list):
        pass
    for name, d, x, y in steps:
        if name == 'INV':
            store['T'][:] = flat(c.f12_inv(unflat(store['F'])))
            continue
        # views with offsets: copy in, run, copy out
        (da, do), (xa, xo), (ya, yo) = arr(d), arr(x), arr(y)
        op = by[name]
        span = 32
        dv, xv, yv = da[do:do + span], xa[xo:xo + span], ya[yo:yo + span]
        if da is xa and do == xo:
            xv = dv
        if da is ya and do == yo:
            yv = dv
        elif xa is ya and xo == yo:
            yv = xv
        exec_op(op, dv, xv, yv)
        da[do:do + len(dv)] = dv


def check_programs():
    """the programs on integers against the oracle: the hard part alone, and whole pairing checks (valid and invalid)"""
    rng = random.Random(2)
    a = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
    t = c.f12_mul(c.f12_conj(a), c.f12_inv(a))
    easy = c.f12_mul(c.f12_frob(t, 2), t)
    lay = Layout(max(o.ntmp for o in OPS) + 1)
    store = {k: [0] * 12 for k in ('F', 'T', 'U', 'W', 'ACC')}
    store['F'] = flat(easy)
    sim_program(prog_final_hard(), store)
    assert unflat(store['T']) == c.final_exponentiation(a), 'hard part program'
    # whole pairing: e(P0, Q0) e(P1, Q1) with the general program, and with pair 1's lines from the fixed-argument model
    sk, h = rng.randrange(1, c.R), rng.randrange(1, c.R)
    Hm = c.E1.mul(c.G1_GEN, h)
    pk = c.E2.mul(c.G2_GEN, sk)
    sig = c.E1.mul(Hm, sk)
    negg2 = c.E2.neg(c.G2_GEN)
    for sgn, want_one in ((sig, True), (c.E1.mul(sig, 2), False)):
        pairs = [(Hm, pk), (sgn, negg2)]
        store = {k: [0] * 12 for k in ('F', 'T', 'U', 'W', 'ACC')}
        store['F'] = flat(c.F12_ONE)
        store['P'] = [Hm[0], Hm[1], sgn[0], sgn[1]]
        store['L'] = [0] * (12 * NSTEPS)
        for pr, (_, q) in enumerate(pairs):
            pt = [0] * 32
            pt[0:6] = [q[0][0], q[0][1], q[1][0], q[1][1], 1, 0]
            pt[6:10] = [q[0][0], q[0][1], q[1][0], q[1][1]]
            store['PT%d' % pr] = pt
        sim_program(PROGRAMS[2][1], store)
        want = c.final_exponentiation(c.miller_loop(pairs))
        assert unflat(store['T']) == want, 'pairing program'
        assert (unflat(store['T']) == c.F12_ONE) == want_one
    return lay


# ------------------------------------------------------------------ emission
def emit(path, lay):
    out = []
    out.append('// GENERATED by tools/gen_wide_tables.py -- do not edit.  Operation tables and programs of the row-wide engine')
    out.append('// (wide_engine.cuh).  Every table and program was checked against the oracle (Fp12 arithmetic, whole pairing checks) by the generator.')
    out.append('#pragma once')
    out.append('// product row: out = (ca0 V[A + a0] + ca1 V[A + a1]) * (cb0 V[B + b0] + cb1 V[B + b1]) -> V[TMP + out]; A, B: the step\'s operand arrays')
    out.append('struct wide_prod { uint8_t a[2], b[2]; int8_t ca[2], cb[2]; uint8_t out, pad[3]; };')
    out.append('// linear row: V[DST + out] = reduce(sum c_i V[idx_i]); idx bit 7 set: operand array A, clear: product scratch')
    out.append('struct wide_lin { uint8_t n, out; uint8_t idx[%d]; int8_t c[%d]; uint8_t pad[2]; };' % (MAXLIN, MAXLIN))
    out.append('struct wide_op { uint16_t prod_off, lin_off; uint8_t nsub, nlin, b_is_const, b_is_a; };')
    prods, lins, ops = [], [], []
    for op in OPS:
        nsub = (len(op.prods) + ROWS - 1) // ROWS
        assert len(op.lins) <= 32 and op.ntmp < 127
        bslots = {i >> 12 for _, b, _ in op.prods for _, i in b}
        aslots = {i >> 12 for a, _, _ in op.prods for _, i in a}
        assert aslots <= {SA} and len(bslots) <= 1 and bslots <= {SA, SB, CONST}, (op.name, aslots, bslots)
        ops.append((op.name, len(prods), len(lins), nsub, len(op.lins), 1 if bslots == {CONST} else 0, 1 if bslots == {SA} else 0, len(op.prods)))
        rows = list(op.prods) + [None] * (nsub * ROWS - len(op.prods))
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
    out.append('#define WOP_INV %d   // interpreter built-in: T <- F^-1 by the lane-local tower code' % len(names))
    names.append('INV')
    out.append('#define WIDE_MAX_TMP %d' % (max(o.ntmp for o in OPS) + 1))
    out.append('// value store (indices of 16-word values)')
    for k, v in lay.base.items():
        out.append('#define WV_%s %d' % (k, v))
    out.append('#define WV_COUNT %d' % lay.count)
    out.append('#define WIDE_STEPS %d' % NSTEPS)
    out.append('#include "wide_rows.cuh"')
    out.append('// programs: two words per step: op | dst << 16,  a | b << 16  (value-store indices of the arrays)')
    for pname, st in PROGRAMS:
        words = []
        for n, d, x, y in st:
            words.append('0x%08xu' % (names.index(n) | lay.ref(d) << 16))
            words.append('0x%08xu' % (lay.ref(x) | lay.ref(y) << 16))
        out.append('#define WIDE_PROG_%s_LEN %d' % (pname, len(st)))
        out.append('BLS_CONST uint32_t WIDE_PROG_%s[%d] = {' % (pname, len(words)))
        for i in range(0, len(words), 8):
            out.append('    ' + ', '.join(words[i:i + 8]) + ',')
        out.append('};')
    out.append('#define WIDE_PROG_MAX %d' % max(len(st) for _, st in PROGRAMS))
    open(path, 'w').write('\n'.join(out) + '\n')
    open(os.path.join(os.path.dirname(path), 'wide_rows.cuh'), 'w').write(
        '// GENERATED by tools/gen_wide_tables.py -- do not edit.\n#pragma once\n'
        '#define WIDE_TABLE_ROWS %d   // DPP rows per product sub-round the engine tables are laid out for (workgroup = 16 x this many threads)\n' % ROWS)


if __name__ == '__main__':
    assert self_check()
    layout = check_programs()
    emit(os.path.join(ROOT, 'agora-blsful_amd', 'csrc', 'wide_tables.cuh'), layout)
    for o in OPS:
        print(o.name, len(o.prods), 'products,', len(o.lins), 'linear rows, longest', max((len(t) for t, _ in o.lins), default=0))
