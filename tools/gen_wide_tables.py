"""Generates agora-blsful_amd/csrc/wide_tables.cuh: the operation tables of the row-wide engine (csrc/wide_engine.cuh).

An operation is a list of PRODUCT sub-rounds -- up to 16 rows each computing  out = (sum ca_i V[a_i]) * (sum cb_i V[b_i])  in
Fp with at most two terms per operand -- followed by one LINEAR phase in which up to 16 rows each write
V[out] = reduce(sum c_i V[idx_i]).  Symbolic value indices are (slot << 12 | offset) with the slots 0 = destination,
1 = first operand, 2 = second operand, 3 = product scratch, 4 = constants; a PROGRAM is a list of (operation, destination
array, operand arrays) steps, e.g. the hard part of the final exponentiation.

The tables are derived here from the tower formulas (Fp12 over the basis w^0..w^5, w^6 = xi = 1 + u, every coefficient an
Fp2 = (re, im)), the Miller-step formulas and the complete addition law.  Self-contained build tooling: the CHECK lives in the
test suite -- tests/wide_tables_check.py simulates the engine's semantics on plain integers (every operation, every program:
the hard part, whole pairing checks valid and invalid, the cut programs, point sums, the cofactor clearing) against the reference
arithmetic there, and tests/test_wide_tables.py also pins the committed header to what this generator writes.  Run:
    python tools/gen_wide_tables.py            (writes csrc/wide_tables.cuh and csrc/wide_rows.cuh)
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab   # the field modulus
DST, SA, SB, TMP, CONST = 0, 1, 2, 3, 4
MAXLIN = 18
ROWS = int(os.environ.get('WIDE_ROWS', '16'))     # DPP rows of the workgroup = products per sub-round (16: four waves, one per SIMD)
X_ABS = 0xd201000000010000       # |x| of the curve (x < 0)
NSTEPS = 68                      # Miller steps: 63 doublings interleaved with 5 additions


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
        terms = [(k, i) for k, i in terms if k]        # an empty list writes zero
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


def op_cyc_sqr(conj=False):
    """Granger-Scott squaring in the cyclotomic subgroup; the Fp4 pairs are (c0, c3), (c1, c4), (c2, c5):
    (a + b s)^2 = (a^2 + xi b^2) + 2ab s;  new = 3 t - 2 z (even coefficients) / 3 t + 2 z (odd).
    conj: the conjugate of the square (odd coefficients negated) -- the last step of a power by the negative x"""
    op = Op('CYC_SQRC' if conj else 'CYC_SQR')
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
            op.lin(scale(merge(scale(t[comp], 3) + [(sgn, z)]), -1 if conj and (k & 1) else 1), idx(DST, 2 * k + comp))
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
# Point workspace of one pair (array PT, 32 values): T = X Y Z (0..5), Q = Xq Yq Zq (6..11), line l0 c2 c3 (12..17), M (18..31).
# The steps are the homogeneous-projective formulas of csrc/pairing.cuh (miller_dbl_step / miller_add_step), cut into levels
# of independent products; the line leaves UNSCALED (c2, c3 still to be multiplied by the G1 point's coordinates: LSCALE).
# Q is PROJECTIVE too (homogeneous; it arrives Jacobian and QPREP converts: no inversion on the key): the add step below is the
# mixed one with theta, lambda scaled by Zq and the line by Zq^2 -- factors in Fp2, which the final exponentiation removes.
PT_T, PT_Q, PT_L, PT_M = 0, 6, 12, 18


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
    lin2(op, sub2(B, E), PT_L + 0)                        # l0 = Y^2 - 3b'Z^2
    lin2(op, S, PT_L + 2, -3)                             # c2 = -3 X^2
    lin2(op, H, PT_L + 4)                                 # c3 = 2YZ = H: PDBL2 reads it from the line (sixteen rows here, one round)
    return op


def op_pdbl2():
    op = Op('PDBL2')
    M = [f2(SA, PT_M + 2 * k) for k in range(5)] + [f2(SA, PT_L + 4)]
    X3 = op.fp2_mul(M[0], M[1])                           # A (B - F)
    Y3 = op.fp2_sqr(M[2])                                 # (B + F)^2
    EE = op.fp2_sqr(M[3])
    Z3 = op.fp2_mul(M[4], M[5])                           # B H
    lin2(op, X3, PT_T + 0, 2)
    lin2(op, sub2(Y3, (scale(EE[0], 12), scale(EE[1], 12))), PT_T + 2)
    lin2(op, Z3, PT_T + 4, 4)
    return op


def op_qprep(stage):
    """Q: Jacobian (X, Y, Z) -> homogeneous (X Z, Y, Z^3).  A: ZZ = Z^2 -> M0, X <- X Z;  B: Z <- Z ZZ;  C: T <- Q"""
    op = Op('QPREP' + stage)
    X, Z = f2(SA, PT_Q), f2(SA, PT_Q + 4)
    if stage == 'A':
        lin2(op, op.fp2_sqr(Z), PT_M + 0)
        lin2(op, op.fp2_mul(X, Z), PT_Q + 0)
    elif stage == 'B':
        lin2(op, op.fp2_mul(Z, f2(SA, PT_M + 0)), PT_Q + 4)
    else:
        for k in range(6):
            op.lin([(1, idx(SA, PT_Q + k))], idx(DST, PT_T + k))
    return op


# add step T <- T + Q, line through T and Q.  With theta = Y Zq - Yq Z, lambda = X Zq - Xq Z (the affine-Q quantities times Zq):
#   cc = theta^2, dd = lambda^2, e = lambda dd, f = (Z Zq) cc, g = (X Zq) dd, h = e + f - 2 g
#   X3 = lambda h,  Y3 = theta (g - h) - e (Y Zq),  Z3 = (Z Zq) e
#   line (times Zq^2): l0 = theta Xq - lambda Yq,  c2 = -theta Zq,  c3 = lambda Zq
# M: 0 theta, 2 lambda, 4 X Zq (then h), 6 Y Zq, 8 Z Zq, 10 cc (then e), 12 dd (then g - h)
def op_padd1():
    op = Op('PADD1')
    X, Y, Z = f2(SA, PT_T), f2(SA, PT_T + 2), f2(SA, PT_T + 4)
    xq, yq, zq = f2(SA, PT_Q), f2(SA, PT_Q + 2), f2(SA, PT_Q + 4)
    yzq, xzq, zzq = op.fp2_mul(Y, zq), op.fp2_mul(X, zq), op.fp2_mul(Z, zq)
    yqz, xqz = op.fp2_mul(yq, Z), op.fp2_mul(xq, Z)
    lin2(op, sub2(yzq, yqz), PT_M + 0)
    lin2(op, sub2(xzq, xqz), PT_M + 2)
    lin2(op, xzq, PT_M + 4)
    lin2(op, yzq, PT_M + 6)
    lin2(op, zzq, PT_M + 8)
    return op


def op_padd2():
    op = Op('PADD2')
    th, la = f2(SA, PT_M + 0), f2(SA, PT_M + 2)
    xq, yq, zq = f2(SA, PT_Q), f2(SA, PT_Q + 2), f2(SA, PT_Q + 4)
    a = op.fp2_mul(th, xq)
    b = op.fp2_mul(la, yq)
    cc = op.fp2_sqr(th)
    dd = op.fp2_sqr(la)
    tz = op.fp2_mul(th, zq)
    lz = op.fp2_mul(la, zq)
    lin2(op, sub2(a, b), PT_L + 0)
    lin2(op, tz, PT_L + 2, -1)
    lin2(op, lz, PT_L + 4)
    lin2(op, cc, PT_M + 10)
    lin2(op, dd, PT_M + 12)
    return op


def op_padd3():
    op = Op('PADD3')
    la, xzq, zzq, cc, dd = f2(SA, PT_M + 2), f2(SA, PT_M + 4), f2(SA, PT_M + 8), f2(SA, PT_M + 10), f2(SA, PT_M + 12)
    e = op.fp2_mul(la, dd)
    f = op.fp2_mul(zzq, cc)
    g = op.fp2_mul(xzq, dd)
    h = sub2(add2(e, f), (scale(g[0], 2), scale(g[1], 2)))
    lin2(op, e, PT_M + 10)
    lin2(op, sub2(g, h), PT_M + 12)
    lin2(op, h, PT_M + 4)
    return op


def op_padd4():
    op = Op('PADD4')
    th, la, h, yzq, zzq, e, gh = (f2(SA, PT_M + o) for o in (0, 2, 4, 6, 8, 10, 12))
    x3 = op.fp2_mul(la, h)
    t = op.fp2_mul(th, gh)
    u = op.fp2_mul(e, yzq)
    z3 = op.fp2_mul(zzq, e)
    lin2(op, x3, PT_T + 0)
    lin2(op, sub2(t, u), PT_T + 2)
    lin2(op, z3, PT_T + 4)
    return op


def op_copy6():
    op = Op('COPY6')                                      # dst[0..5] = the line of the point workspace
    for k in range(6):
        op.lin([(1, idx(SA, PT_L + k))], idx(DST, k))
    return op


def v1(off):
    return (1, idx(SA, off))


def op_lscale(pairs, nsteps=1):
    """the lines of one step (array L[step]: pair p at 6p) evaluated at the pairs' G1 points, which are JACOBIAN (X, Y, Z): the
    line l0 + c2 x w^2 + c3 y w^3 times Z^3 (a factor in Fp, which the final exponentiation removes) is
    l0 Z^3 + c2 (X Z) w^2 + c3 Y w^3; array P holds (X Z, Y, Z^3, -) of pair p at 4p (PPREPA / PPREPB).  No inversion anywhere.
    nsteps > 1: the same for that many consecutive steps in one operation (the lines of a Miller loop are all known before it
    starts: evaluating them eight steps at a time fills the sixteen rows and saves sixty program steps)."""
    op = Op('LSCALE' + ''.join(str(p) for p in pairs) + ('X%d' % nsteps if nsteps > 1 else ''))
    for st in range(nsteps):
        for pr in pairs:
            for part, pc in ((0, 2), (2, 0), (4, 1)):         # l0 * Z^3, c2 * XZ, c3 * Y
                for comp in range(2):
                    t = op.prod([(1, idx(SA, 12 * st + 6 * pr + part + comp))], [(1, idx(SB, 4 * pr + pc))])
                    op.lin([(1, t)], idx(DST, 12 * st + 6 * pr + part + comp))
    return op


def op_pprep(stage, pairs):
    """array P, pair p at 4p: (X, Y, Z, -) -> A: (X Z, Y, Z, Z^2) -> B: (X Z, Y, Z^3, Z^2)"""
    op = Op('PPREP%s%s' % (stage, ''.join(str(p) for p in pairs)))
    for pr in pairs:
        o = 4 * pr
        if stage == 'A':
            op.lin([(1, op.prod([v1(o + 2)], [v1(o + 2)]))], idx(DST, o + 3))
            op.lin([(1, op.prod([v1(o)], [v1(o + 2)]))], idx(DST, o))
        else:
            op.lin([(1, op.prod([v1(o + 2)], [v1(o + 3)]))], idx(DST, o + 2))
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


def op_f6inv(stage):
    """the inversion of N = n0 + n1 v + n2 v^2 in Fp6 (v = w^2; N sits at the even coefficients of an Fp12 array, its odd
    coefficients are free) cut into table operations around ONE inversion in Fp (the interpreter's built-in FPINV):
      1: t0 = n0^2 - xi n1 n2, t1 = xi n2^2 - n0 n1, t2 = n1^2 - n0 n2        -> odd coefficients 1, 3, 5 of the array
      2: d = n0 t0 + xi (n2 t1 + n1 t2)                                       -> dst values 0, 1
      3: s = d0^2 + d1^2  (the norm of d)                                     -> value 2 of the same array
      4: d^-1 = (d0, -d1) s^-1  with s^-1 at value 3                          -> values 4, 5
      5: N^-1 = (t0, t1, t2) d^-1 -> even coefficients, zeros -> odd ones     (a = the Fp12 array, b = the array of d^-1)"""
    op = Op('F6INV%d' % stage)
    n = [coef(SA, 0), coef(SA, 2), coef(SA, 4)]
    t = [coef(SA, 1), coef(SA, 3), coef(SA, 5)]
    if stage == 1:
        n0n0, n1n1, n2n2 = op.fp2_sqr(n[0]), op.fp2_sqr(n[1]), op.fp2_sqr(n[2])
        n1n2, n0n1, n0n2 = op.fp2_mul(n[1], n[2]), op.fp2_mul(n[0], n[1]), op.fp2_mul(n[0], n[2])
        lin2(op, sub2(n0n0, mul_xi(*n1n2)), 2)
        lin2(op, sub2(mul_xi(*n2n2), n0n1), 6)
        lin2(op, sub2(n1n1, n0n2), 10)
    elif stage == 2:
        a, b, cc = op.fp2_mul(n[0], t[0]), op.fp2_mul(n[2], t[1]), op.fp2_mul(n[1], t[2])
        lin2(op, add2(a, mul_xi(*add2(b, cc))), 0)
    elif stage == 3:
        op.lin([(1, op.prod([v1(0)], [v1(0)])), (1, op.prod([v1(1)], [v1(1)]))], idx(DST, 2))
    elif stage == 4:
        op.lin([(1, op.prod([v1(0)], [v1(3)]))], idx(DST, 4))
        op.lin([(-1, op.prod([v1(1)], [v1(3)]))], idx(DST, 5))
    else:
        di = ((1, idx(SB, 4)), (1, idx(SB, 5)))
        for k in range(3):
            lin2(op, op.fp2_mul(t[k], di), 4 * k)
            op.lin([], idx(DST, 4 * k + 2))
            op.lin([], idx(DST, 4 * k + 3))
    return op


OPS += [op_f6inv(k) for k in (1, 2, 3, 4, 5)]
OPS += [op_pdbl1(), op_pdbl2(), op_padd1(), op_padd2(), op_padd3(), op_padd4(), op_copy6(), op_lscale((0, 1)), op_mul_line(),
        op_lscale((0,)), op_lscale((1,)), op_lscale((0,), 8), op_lscale((0,), 4), op_lscale((1,), 8), op_lscale((1,), 4)] + [op_pprep(st, pr) for pr in ((0, 1), (0,), (1,)) for st in 'AB'] + [op_qprep(st) for st in 'ABC']
OPS += [op_cyc_sqr(True)]



# ------------------------------------------------------------------ point sums on the engine (table set PT)
# Complete addition on y^2 = x^3 + b in homogeneous projective coordinates (Renes, Costello, Batina 2016, algorithm 7 for a = 0).
# E(Fp) and E'(Fp2) have odd order, so the formulas have NO exceptional inputs: doubling, opposite points and the identity
# (0 : 1 : 0) go through the same twelve products -- which is what a table-driven engine without branches needs.  Two levels
# of six Fp2 (Fp) products each:
#   t0 = X1 X2, t1 = Y1 Y2, t2 = Z1 Z2, t3 = X1 Y2 + X2 Y1, t4 = Y1 Z2 + Y2 Z1, t5 = X1 Z2 + X2 Z1
#   m0 = 3 t0, m1 = t1 - 3b t2, m2 = t1 + 3b t2, m3 = t3, m4 = t4, m5 = 3b t5
#   X3 = m3 m1 - m4 m5,  Y3 = m1 m2 + m5 m0,  Z3 = m2 m4 + m0 m3
# A PAIR SLOT holds the two operands and the m values: G2 [P1 (6) | P2 (6) | m (12)] = 24 values, G1 [3 | 3 | 6] = 12.  The
# X<k> variants run k additions of consecutive slots in one step (so that the sixteen rows of a sub-round stay busy); the
# second level writes sum j to operand j % 2 of slot j // 2 of the destination array: the next level of the tree.
# Jacobian <-> homogeneous at the ends: (X, Y, Z) -> (X Z : Y : Z^3) and (X : Y : Z) -> (X Z, Y Z^2, Z).
G2S, G1S = 24, 12


def fp2_mul_sum(op, x1, x2, y1, y2):
    """(x1 + x2)(y1 + y2) on single-term Fp2 components: schoolbook over two-term operand sums (four products)"""
    rr = op.prod([x1[0], x2[0]], [y1[0], y2[0]])
    ii = op.prod([x1[1], x2[1]], [y1[1], y2[1]])
    ri = op.prod([x1[0], x2[0]], [y1[1], y2[1]])
    ir = op.prod([x1[1], x2[1]], [y1[0], y2[0]])
    return ([(1, rr), (-1, ii)], [(1, ri), (1, ir)])


def b3_g2(v):
    x = mul_xi(*v)                                          # 3 b' = 12 (1 + u)
    return (scale(x[0], 12), scale(x[1], 12))


def op_c2add1(m):
    op = Op('C2ADD1X%d' % m)
    for k in range(m):
        o = G2S * k
        X1, Y1, Z1, X2, Y2, Z2 = (f2(SA, o + 2 * j) for j in range(6))
        t0, t1, t2 = op.fp2_mul(X1, X2), op.fp2_mul(Y1, Y2), op.fp2_mul(Z1, Z2)
        t3 = sub2(sub2(fp2_mul_sum(op, X1, Y1, X2, Y2), t0), t1)
        t4 = sub2(sub2(fp2_mul_sum(op, Y1, Z1, Y2, Z2), t1), t2)
        t5 = sub2(sub2(fp2_mul_sum(op, X1, Z1, X2, Z2), t0), t2)
        bt2 = b3_g2(t2)
        lin2(op, t0, o + 12, 3)
        lin2(op, sub2(t1, bt2), o + 14)
        lin2(op, add2(t1, bt2), o + 16)
        lin2(op, t3, o + 18)
        lin2(op, t4, o + 20)
        lin2(op, b3_g2(t5), o + 22)
    return op


def op_c2add2(m):
    op = Op('C2ADD2X%d' % m)
    for k in range(m):
        o = G2S * k + 12
        m0, m1, m2, m3, m4, m5 = (f2(SA, o + 2 * j) for j in range(6))
        d = G2S * (k // 2) + 6 * (k % 2)
        lin2(op, sub2(op.fp2_mul(m3, m1), op.fp2_mul(m4, m5)), d)
        lin2(op, add2(op.fp2_mul(m1, m2), op.fp2_mul(m5, m0)), d + 2)
        lin2(op, add2(op.fp2_mul(m2, m4), op.fp2_mul(m0, m3)), d + 4)
    return op


def g2_point_off(i):
    return G2S * (i // 2) + 6 * (i % 2)


def op_c2j2h(stage, m):
    """Jacobian -> homogeneous for the m points of m / 2 consecutive slots: A: ZZ = Z^2 (into the slot's m area), X <- X Z;  B: Z <- Z ZZ"""
    op = Op('C2J2H%sX%d' % (stage, m))
    for i in range(m):
        o, zz = g2_point_off(i), G2S * (i // 2) + 12 + 2 * (i % 2)
        if stage == 'A':
            lin2(op, op.fp2_sqr(f2(SA, o + 4)), zz)
            lin2(op, op.fp2_mul(f2(SA, o), f2(SA, o + 4)), o)
        else:
            lin2(op, op.fp2_mul(f2(SA, o + 4), f2(SA, zz)), o + 4)
    return op


def op_c2h2j(stage):
    """homogeneous -> Jacobian of the point at operand 0 of a slot: A: ZZ = Z^2, X <- X Z;  B: Y <- Y ZZ"""
    op = Op('C2H2J%s' % stage)
    if stage == 'A':
        lin2(op, op.fp2_sqr(f2(SA, 4)), 12)
        lin2(op, op.fp2_mul(f2(SA, 0), f2(SA, 4)), 0)
    else:
        lin2(op, op.fp2_mul(f2(SA, 2), f2(SA, 12)), 2)
    return op


def op_c1add1(m):
    op = Op('C1ADD1X%d' % m)
    for k in range(m):
        o = G1S * k
        X1, Y1, Z1, X2, Y2, Z2 = (v1(o + j) for j in range(6))
        t0, t1, t2 = op.prod([X1], [X2]), op.prod([Y1], [Y2]), op.prod([Z1], [Z2])
        T3, T4, T5 = op.prod([X1, Y1], [X2, Y2]), op.prod([Y1, Z1], [Y2, Z2]), op.prod([X1, Z1], [X2, Z2])
        op.lin([(3, t0)], idx(DST, o + 6))
        op.lin([(1, t1), (-12, t2)], idx(DST, o + 7))
        op.lin([(1, t1), (12, t2)], idx(DST, o + 8))
        op.lin([(1, T3), (-1, t0), (-1, t1)], idx(DST, o + 9))
        op.lin([(1, T4), (-1, t1), (-1, t2)], idx(DST, o + 10))
        op.lin([(12, T5), (-12, t0), (-12, t2)], idx(DST, o + 11))
    return op


def op_c1add2(m):
    op = Op('C1ADD2X%d' % m)
    for k in range(m):
        o = G1S * k + 6
        m0, m1, m2, m3, m4, m5 = (v1(o + j) for j in range(6))
        d = G1S * (k // 2) + 3 * (k % 2)
        op.lin([(1, op.prod([m3], [m1])), (-1, op.prod([m4], [m5]))], idx(DST, d))
        op.lin([(1, op.prod([m1], [m2])), (1, op.prod([m5], [m0]))], idx(DST, d + 1))
        op.lin([(1, op.prod([m2], [m4])), (1, op.prod([m0], [m3]))], idx(DST, d + 2))
    return op


def g1_point_off(i):
    return G1S * (i // 2) + 3 * (i % 2)


def op_c1j2h(stage, m):
    op = Op('C1J2H%sX%d' % (stage, m))
    for i in range(m):
        o, zz = g1_point_off(i), G1S * (i // 2) + 6 + (i % 2)
        if stage == 'A':
            op.lin([(1, op.prod([v1(o + 2)], [v1(o + 2)]))], idx(DST, zz))
            op.lin([(1, op.prod([v1(o)], [v1(o + 2)]))], idx(DST, o))
        else:
            op.lin([(1, op.prod([v1(o + 2)], [v1(zz)]))], idx(DST, o + 2))
    return op


def op_c1h2j(stage):
    op = Op('C1H2J%s' % stage)
    if stage == 'A':
        op.lin([(1, op.prod([v1(2)], [v1(2)]))], idx(DST, 6))
        op.lin([(1, op.prod([v1(0)], [v1(2)]))], idx(DST, 0))
    else:
        op.lin([(1, op.prod([v1(1)], [v1(6)]))], idx(DST, 1))
    return op


# ---- general-register forms for straight-line point programs (the cofactor clearing of hash-to-G2): a REGISTER is a pair slot whose
# first operand holds the point; an addition takes its operands from two registers (array A and array B of the step) and
# leaves the m values in the scratch register, the second level turns them into a point anywhere
def op_c2gadd1():
    op = Op('C2GADD1')
    X1, Y1, Z1 = (f2(SA, 2 * j) for j in range(3))
    X2, Y2, Z2 = (f2(SB, 2 * j) for j in range(3))
    t0, t1, t2 = op.fp2_mul(X1, X2), op.fp2_mul(Y1, Y2), op.fp2_mul(Z1, Z2)
    t3 = sub2(sub2(fp2_mul_sum(op, X1, Y1, X2, Y2), t0), t1)
    t4 = sub2(sub2(fp2_mul_sum(op, Y1, Z1, Y2, Z2), t1), t2)
    t5 = sub2(sub2(fp2_mul_sum(op, X1, Z1, X2, Z2), t0), t2)
    bt2 = b3_g2(t2)
    lin2(op, t0, 0, 3)
    lin2(op, sub2(t1, bt2), 2)
    lin2(op, add2(t1, bt2), 4)
    lin2(op, t3, 6)
    lin2(op, t4, 8)
    lin2(op, b3_g2(t5), 10)
    return op


def op_c2gadd2():
    op = Op('C2GADD2')
    m0, m1, m2, m3, m4, m5 = (f2(SA, 2 * j) for j in range(6))
    lin2(op, sub2(op.fp2_mul(m3, m1), op.fp2_mul(m4, m5)), 0)
    lin2(op, add2(op.fp2_mul(m1, m2), op.fp2_mul(m5, m0)), 2)
    lin2(op, add2(op.fp2_mul(m2, m4), op.fp2_mul(m0, m3)), 4)
    return op


def scale2(v, k):
    return (scale(v[0], k), scale(v[1], k))


def op_c2dbl1():
    """doubling on E'(Fp2) (Renes-Costello-Batina 2016, algorithm 9 for a = 0), complete like the addition; first half:
    A = Y^2, B = Y Z, C = 3b' Z^2, D = X Y  ->  MS = (A, B, C, D, E = A - 3C, F = A + C): ten products, one sub-round"""
    op = Op('C2DBL1')
    X, Y, Z = (f2(SA, 2 * j) for j in range(3))
    A, B, C, D = op.fp2_sqr(Y), op.fp2_mul(Y, Z), b3_g2(op.fp2_sqr(Z)), op.fp2_mul(X, Y)
    for j, v in enumerate((A, B, C, D, sub2(A, scale2(C, 3)), add2(A, C))):
        lin2(op, v, 2 * j)
    return op


def op_c2dbl2():
    """second half: X3 = 2 E D, Y3 = E F + 8 A C, Z3 = 8 A B (twelve products, one sub-round)"""
    op = Op('C2DBL2')
    A, B, C, D, E, F = (f2(SA, 2 * j) for j in range(6))
    lin2(op, op.fp2_mul(E, D), 0, 2)
    lin2(op, add2(op.fp2_mul(E, F), scale2(op.fp2_mul(A, C), 8)), 2)
    lin2(op, op.fp2_mul(A, B), 4, 8)
    return op


def op_c2neg():
    op = Op('C2NEG')
    for k in range(6):
        op.lin([(-1 if k in (2, 3) else 1, idx(SA, k))], idx(DST, k))
    return op


def op_c2psi():
    """psi on homogeneous coordinates: (X : Y : Z) -> (cx conj X : cy conj Y : conj Z); constants cx, cy at CONST 0..3"""
    op = Op('C2PSI')
    cx, cy = f2(CONST, 0), f2(CONST, 2)
    X = ((1, idx(SA, 0)), (-1, idx(SA, 1)))
    Y = ((1, idx(SA, 2)), (-1, idx(SA, 3)))
    lin2(op, op.fp2_mul(X, cx), 0)
    lin2(op, op.fp2_mul(Y, cy), 2)
    op.lin([(1, idx(SA, 4))], idx(DST, 4))
    op.lin([(-1, idx(SA, 5))], idx(DST, 5))
    return op


def op_c1gadd1():
    op = Op('C1GADD1')
    X1, Y1, Z1 = (v1(j) for j in range(3))
    X2, Y2, Z2 = ((1, idx(SB, j)) for j in range(3))
    t0, t1, t2 = op.prod([X1], [X2]), op.prod([Y1], [Y2]), op.prod([Z1], [Z2])
    T3, T4, T5 = op.prod([X1, Y1], [X2, Y2]), op.prod([Y1, Z1], [Y2, Z2]), op.prod([X1, Z1], [X2, Z2])
    op.lin([(3, t0)], idx(DST, 0))
    op.lin([(1, t1), (-12, t2)], idx(DST, 1))
    op.lin([(1, t1), (12, t2)], idx(DST, 2))
    op.lin([(1, T3), (-1, t0), (-1, t1)], idx(DST, 3))
    op.lin([(1, T4), (-1, t1), (-1, t2)], idx(DST, 4))
    op.lin([(12, T5), (-12, t0), (-12, t2)], idx(DST, 5))
    return op


def op_c1gadd2():
    op = Op('C1GADD2')
    m0, m1, m2, m3, m4, m5 = (v1(j) for j in range(6))
    op.lin([(1, op.prod([m3], [m1])), (-1, op.prod([m4], [m5]))], idx(DST, 0))
    op.lin([(1, op.prod([m1], [m2])), (1, op.prod([m5], [m0]))], idx(DST, 1))
    op.lin([(1, op.prod([m2], [m4])), (1, op.prod([m0], [m3]))], idx(DST, 2))
    return op


# ---- the 11-isogeny of hash-to-G1 for the two mapped points at once (array ISO: map m at ISO_STRIDE m).  The SSWU map leaves
# x' = xn / xd and y; the rational map x = xnum(x') / xden(x'), y = y ynum(x') / yden(x') is evaluated on homogenised polynomials
# (sum_i k_i xn^i xd^(deg - i)): powers of xn and xd in four levels, the 33 mixed monomials in one step, then every polynomial is ONE
# linear row over (monomial x constant) products -- a single reduction per polynomial, where a Horner chain is ~100 dependent
# multiplications -- and two steps assemble the homogeneous point (XN YD : y YN zx : zx YD), zx = XD xd.  Constants: array
# CONST of table set PT, ISO_K + (0, 12, 23, 39) for xnum (12), xden (11), ynum (16), yden (16), coefficient of x^i at i.
ISO_STRIDE, ISO_XN, ISO_XD, ISO_Y, ISO_POW_N, ISO_POW_D, ISO_MONO, ISO_POLY, ISO_ZX, ISO_B, ISO_PT = 80, 0, 1, 2, 3, 17, 31, 64, 68, 70, 73
ISO_K = 4
ISO_KOFF = {'xnum': ISO_K, 'xden': ISO_K + 12, 'ynum': ISO_K + 23, 'yden': ISO_K + 39}
ISO_DEG = {'xnum': 11, 'xden': 10, 'ynum': 15, 'yden': 15}


def iso_pow(m, var, k):
    """value index of var^k (k >= 1) of map m"""
    base = ISO_STRIDE * m
    if k == 1:
        return base + (ISO_XN if var == 'n' else ISO_XD)
    return base + (ISO_POW_N if var == 'n' else ISO_POW_D) + k - 2


def iso_mono(m, deg, i):
    """value index of xn^i xd^(deg - i) of map m (the pure powers live in the power tables)"""
    if i == 0:
        return iso_pow(m, 'd', deg)
    if i == deg:
        return iso_pow(m, 'n', deg)
    off = {11: 0, 10: 10, 15: 19}[deg]
    return ISO_STRIDE * m + ISO_MONO + off + i - 1


def op_iso_pow(level):
    op = Op('C1ISOP%d' % level)
    lo, hi, mul = {1: (2, 2, 1), 2: (3, 4, 2), 3: (5, 8, 4), 4: (9, 15, 8)}[level]
    for m in range(2):
        for var in 'nd':
            for k in range(lo, hi + 1):
                t = op.prod([(1, idx(SA, iso_pow(m, var, mul)))], [(1, idx(SA, iso_pow(m, var, k - mul)))])
                op.lin([(1, t)], idx(DST, iso_pow(m, var, k)))
    return op


def op_iso_mono():
    op = Op('C1ISOM')
    for m in range(2):
        for deg in (11, 10, 15):
            for i in range(1, deg):
                t = op.prod([(1, idx(SA, iso_pow(m, 'n', i)))], [(1, idx(SA, iso_pow(m, 'd', deg - i)))])
                op.lin([(1, t)], idx(DST, iso_mono(m, deg, i)))
    return op


def op_iso_poly():
    op = Op('C1ISOK')
    for m in range(2):
        for j, name in enumerate(('xnum', 'xden', 'ynum', 'yden')):
            deg, monic = ISO_DEG[name], name in ('xden', 'yden')
            terms = []
            for i in range(deg + 1):
                if monic and i == deg:
                    terms.append((1, idx(SA, iso_mono(m, deg, i))))          # leading coefficient 1: the monomial itself
                else:
                    terms.append((1, op.prod([(1, idx(SA, iso_mono(m, deg, i)))], [(1, idx(CONST, ISO_KOFF[name] + i))])))
            op.lin(terms, idx(DST, ISO_STRIDE * m + ISO_POLY + j))
    return op


def op_iso_asm(stage):
    op = Op('C1ISOA%d' % stage)
    for m in range(2):
        b = ISO_STRIDE * m
        XN, XD, YN, YD = (b + ISO_POLY + j for j in range(4))
        if stage == 1:
            op.lin([(1, op.prod([(1, idx(SA, XD))], [(1, idx(SA, b + ISO_XD))]))], idx(DST, b + ISO_ZX))
            op.lin([(1, op.prod([(1, idx(SA, XN))], [(1, idx(SA, YD))]))], idx(DST, b + ISO_PT))
            op.lin([(1, op.prod([(1, idx(SA, b + ISO_Y))], [(1, idx(SA, YN))]))], idx(DST, b + ISO_B))
        else:
            op.lin([(1, op.prod([(1, idx(SA, b + ISO_B))], [(1, idx(SA, b + ISO_ZX))]))], idx(DST, b + ISO_PT + 1))
            op.lin([(1, op.prod([(1, idx(SA, b + ISO_ZX))], [(1, idx(SA, YD))]))], idx(DST, b + ISO_PT + 2))
    return op


OPS_PT = ([op_c2add1(m) for m in (4, 2, 1)] + [op_c2add2(m) for m in (4, 2, 1)] + [op_c2j2h('A', 8), op_c2j2h('B', 8), op_c2h2j('A'), op_c2h2j('B')] +
          [op_c1add1(m) for m in (8, 4, 2, 1)] + [op_c1add2(m) for m in (8, 4, 2, 1)] + [op_c1j2h('A', 8), op_c1j2h('B', 8), op_c1h2j('A'), op_c1h2j('B')] +
          [op_c2gadd1(), op_c2gadd2(), op_c2neg(), op_c2psi(), op_c2j2h('A', 1), op_c2j2h('B', 1),
           op_c1gadd1(), op_c1gadd2(), op_c1j2h('A', 1), op_c1j2h('B', 1)] +
          [op_iso_pow(k) for k in (1, 2, 3, 4)] + [op_iso_mono(), op_iso_poly(), op_iso_asm(1), op_iso_asm(2)] + [op_c2dbl1(), op_c2dbl2()])
PT_POINTS = 16                     # points per workgroup: a tree of four levels


def prog_point_tree(g, jac_in, jac_out):
    """the sum of sixteen points of group g (1: G1, 2: G2): L0 (eight pair slots) -> L1 -> L2 -> L3 -> operand 0 of L4"""
    S = G2S if g == 2 else G1S
    c = 'C%d' % g
    st = []
    if jac_in:
        for grp in range(2):
            a = ('L0', 4 * S * grp)
            st += [(c + 'J2HAX8', a, a, a), (c + 'J2HBX8', a, a, a)]
    if g == 2:
        for grp in range(2):
            a = ('L0', 4 * S * grp)
            st += [(c + 'ADD1X4', a, a, a), (c + 'ADD2X4', ('L1', 2 * S * grp), a, a)]
        st += [(c + 'ADD1X4', 'L1', 'L1', 'L1'), (c + 'ADD2X4', 'L2', 'L1', 'L1')]
    else:
        st += [(c + 'ADD1X8', 'L0', 'L0', 'L0'), (c + 'ADD2X8', 'L1', 'L0', 'L0'), (c + 'ADD1X4', 'L1', 'L1', 'L1'), (c + 'ADD2X4', 'L2', 'L1', 'L1')]
    st += [(c + 'ADD1X2', 'L2', 'L2', 'L2'), (c + 'ADD2X2', 'L3', 'L2', 'L2'), (c + 'ADD1X1', 'L3', 'L3', 'L3'), (c + 'ADD2X1', 'L4', 'L3', 'L3')]
    if jac_out:
        st += [(c + 'H2JA', 'L4', 'L4', 'L4'), (c + 'H2JB', 'L4', 'L4', 'L4')]
    return st


def prog_g2_clear_cofactor():
    """h_eff P on G2 (RFC 9380 G.3, Budroni-Pintore, as oracle g2_clear_cofactor) over registers R0..R4 (R0 = P, Jacobian in,
    Jacobian out in R3): x = -|x|, so [x]Q = -[|x|]Q by double-and-add from the top bit, every point operation a complete addition"""
    def add(d, a, b):
        if a == b:                              # a doubling: 10 + 12 products in two sub-rounds instead of 21 + 18 in four
            return [('C2DBL1', 'MS', a, a), ('C2DBL2', d, 'MS', 'MS')]
        return [('C2GADD1', 'MS', a, b), ('C2GADD2', d, 'MS', 'MS')]

    def mul_x(d, base, acc):                    # d <- [x] base, acc: a work register (neither d's operand base nor ... d may be acc)
        st = []
        for i in range(62, -1, -1):             # bit 63 is set: the accumulator starts as base; the first doubling reads base itself
            st += add(acc, base, base) if i == 62 else add(acc, acc, acc)
            if (X_ABS >> i) & 1:
                st += add(acc, acc, base)
        return st + [('C2NEG', d, acc, acc)]

    st = [('C2J2HAX1', 'R0', 'R0', 'R0'), ('C2J2HBX1', 'R0', 'R0', 'R0')]
    st += mul_x('R1', 'R0', 'R4')                                         # t1 = [x] P
    st += [('C2PSI', 'R2', 'R0', 'R0')]                                   # t2 = psi(P)
    st += add('R4', 'R0', 'R0') + [('C2PSI', 'R3', 'R4', 'R4'), ('C2PSI', 'R3', 'R3', 'R3')]     # t3 = psi^2(2P)
    st += [('C2NEG', 'R4', 'R2', 'R2')] + add('R3', 'R3', 'R4')           # t3 -= t2
    st += add('R2', 'R1', 'R2')                                           # t2 = t1 + t2
    st += mul_x('R2', 'R2', 'R4')                                         # t2 = [x] t2   (acc = R4, base R2 is read until the last step)
    st += add('R3', 'R3', 'R2')                                           # t3 += t2
    st += [('C2NEG', 'R4', 'R1', 'R1')] + add('R3', 'R3', 'R4')           # t3 -= t1
    st += [('C2NEG', 'R4', 'R0', 'R0')] + add('R3', 'R3', 'R4')           # t3 -= P
    st += [('C2H2JA', 'R3', 'R3', 'R3'), ('C2H2JB', 'R3', 'R3', 'R3')]
    return st


PROGRAMS_PT = [('G%d_%s%s' % (g, 'J' if ji else 'H', 'J' if jo else 'H'), prog_point_tree(g, ji, jo)) for g in (1, 2) for ji in (1, 0) for jo in (1, 0)]
def prog_g1_clear_cofactor():
    """h_eff P = (1 - x) P = P + |x| P on G1 (RFC 9380 8.8.1): double-and-add from the top bit, complete additions; registers as
    for G2 (R0 = P Jacobian in, R1 = the result, Jacobian out)"""
    def add(d, a, b):
        return [('C1GADD1', 'MS', a, b), ('C1GADD2', d, 'MS', 'MS')]

    st = [('C1J2HAX1', 'R0', 'R0', 'R0'), ('C1J2HBX1', 'R0', 'R0', 'R0')]
    for i in range(62, -1, -1):
        st += add('R1', 'R0', 'R0') if i == 62 else add('R1', 'R1', 'R1')
        if (X_ABS >> i) & 1:
            st += add('R1', 'R1', 'R0')
    st += add('R1', 'R1', 'R0')
    return st + [('C1H2JA', 'R1', 'R1', 'R1'), ('C1H2JB', 'R1', 'R1', 'R1')]


def prog_g1_hash_tail():
    """everything of hash-to-G1 behind the two SSWU maps: the isogeny of both points, their sum, the cofactor clearing; result
    (Jacobian) in R1"""
    st = [('C1ISOP%d' % k, 'ISO', 'ISO', 'ISO') for k in (1, 2, 3, 4)]
    st += [('C1ISOM', 'ISO', 'ISO', 'ISO'), ('C1ISOK', 'ISO', 'ISO', 'ISO'), ('C1ISOA1', 'ISO', 'ISO', 'ISO'), ('C1ISOA2', 'ISO', 'ISO', 'ISO')]
    st += [('C1GADD1', 'MS', ('ISO', ISO_PT), ('ISO', ISO_STRIDE + ISO_PT)), ('C1GADD2', 'R0', 'MS', 'MS')]
    return st + prog_g1_clear_cofactor()[2:]        # R0 is homogeneous already: skip the Jacobian -> homogeneous steps


def prog_g2_hash_tail():
    """hash-to-G2 behind the two SSWU maps and their isogenies: R0, R1 = the two points of E2 (homogeneous); their sum, then the
    cofactor clearing; result (Jacobian) in R3"""
    return [('C2GADD1', 'MS', 'R0', 'R1'), ('C2GADD2', 'R0', 'MS', 'MS')] + prog_g2_clear_cofactor()[2:]


def prog_g1_hash_map():
    """the same WITHOUT the cofactor clearing: the isogeny of both points and their sum, Jacobian in R0 -- what a Bls12381G1Impl
    verification pairs with the key when its second pair is (sig, -[c] g2) (csrc/g2neg_lines.cuh)"""
    return prog_g1_hash_tail()[:10] + [('C1H2JA', 'R0', 'R0', 'R0'), ('C1H2JB', 'R0', 'R0', 'R0')]


PROGRAMS_PT += [('G2_CLEAR', prog_g2_clear_cofactor()), ('G1_CLEAR', prog_g1_clear_cofactor()), ('G1_HASH_TAIL', prog_g1_hash_tail()),
                ('G2_HASH_TAIL', prog_g2_hash_tail()), ('G1_HASH_MAP', prog_g1_hash_map())]


# ------------------------------------------------------------------ programs: sequences of (op, dst, a, b) over the value store


class Layout:
    """value-store indices (16-word values): named arrays in order, the product scratch 'TMP' among them"""
    def __init__(self, arrays):
        self.base = {}
        off = 0
        for name, n in arrays:
            self.base[name] = off
            off += n
        self.count = off

    def ref(self, r):
        """'F' or ('L', 12) -> value index; a plain integer (the argument of a built-in step) stays what it is"""
        if isinstance(r, int):
            return r
        if isinstance(r, tuple):
            return self.base[r[0]] + r[1]
        return self.base[r]


def prog_pow_x(dst, a):
    """dst = a^x (x < 0) in the cyclotomic subgroup: the first squaring reads a, the last one conjugates (|x| is even) and writes dst"""
    assert X_ABS % 2 == 0 and a != 'ACC'
    st = []
    for i in range(62, -1, -1):
        st.append(('CYC_SQRC' if i == 0 else 'CYC_SQR', dst if i == 0 else 'ACC', a if i == 62 else 'ACC', 'ACC'))
        if (X_ABS >> i) & 1:
            st.append(('MUL', 'ACC', 'ACC', a))
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


# hand-overs of the streamed cut (PUB / ACQ below): line steps [b[j-1], b[j]) travel together.  The producer makes a line in ~3.2 us,
# the consumer uses one in ~4.4 us, both start together: a short first chunk gets the consumer going, later ones may grow as the
# producer pulls ahead (3.2 b[j] <= 3.2 b[0] + 4.4 b[j-1]).  Multiples of four: LSCALE<p>X4 / X8 evaluate four / eight steps at once.
# The Miller loop itself runs on TWO consumers (the split of prog_post_hi / prog_post_lo below, at SPLIT_S): the second one starts at
# line step 40 and gets short chunks again from there
STREAM_BOUNDS = [4, 8, 12, 20, 32, 40, 44, 52, 60, 68]
SPLIT_S = 27                     # the streamed cut's own split (below): its second consumer starts when line step 40 exists
assert STREAM_BOUNDS[-1] == NSTEPS and all(b % 4 == 0 for b in STREAM_BOUNDS)


def prog_key_lines(pair, publish=False):
    """the UNSCALED line coefficients of the Miller steps of pair `pair` (its point workspace holds Q, Jacobian) -> L[step][pair].
    publish: when a chunk of STREAM_BOUNDS is complete the built-in PUB (first step, steps) hands it to the workgroup that runs
    prog_miller_stream beside this one (kernels.cuh k_pairing_stream)"""
    pt = 'PT%d' % pair
    st, step = [('QPREP' + x, pt, pt, pt) for x in 'ABC'], 0

    def done():
        if publish and step in STREAM_BOUNDS:
            j = STREAM_BOUNDS.index(step)
            first = STREAM_BOUNDS[j - 1] if j else 0
            st.append(('PUB', first, step - first, 0))
    for i in range(62, -1, -1):
        st += [('PDBL1', pt, pt, pt), ('PDBL2', pt, pt, pt), ('COPY6', ('L', 12 * step + 6 * pair), pt, pt)]
        step += 1
        done()
        if (X_ABS >> i) & 1:
            st += [('PADD%d' % k, pt, pt, pt) for k in (1, 2, 3, 4)] + [('COPY6', ('L', 12 * step + 6 * pair), pt, pt)]
            step += 1
            done()
    assert step == NSTEPS
    return st


def prog_pprep(pairs):
    sfx = ''.join(str(p) for p in pairs)
    return [('PPREPA' + sfx, 'P', 'P', 'P'), ('PPREPB' + sfx, 'P', 'P', 'P')]


def prog_miller(pairs=(0, 1)):
    """F <- F * prod over `pairs` of the pairs' Miller functions (not yet conjugated), the lines taken from L and evaluated here at
    the pairs' G1 points (array P, prepared by prog_pprep): for a single pair all 68 lines up front, eight steps per operation"""
    sfx = ''.join(str(p) for p in pairs)
    st, step = [], 0
    packed = len(pairs) == 1
    if packed:
        for s0 in range(0, NSTEPS, 8):
            k = min(8, NSTEPS - s0)
            assert k in (8, 4)
            ls = ('L', 12 * s0)
            st.append(('LSCALE%sX%d' % (sfx, k), ls, ls, 'P'))
    for i in range(62, -1, -1):
        if i != 62:
            st.append(('SQR', 'F', 'F', 'F'))
        for _ in range(2 if (X_ABS >> i) & 1 else 1):
            ls = ('L', 12 * step)
            if not packed:
                st.append(('LSCALE' + sfx, ls, ls, 'P'))
            st += [('MUL_LINE', 'F', 'F', ('L', 12 * step + 6 * p)) for p in pairs]
            step += 1
    return st


def miller_line_steps(i_from, i_to):
    """(first line step, one past the last) of iterations i_from..i_to (downwards)"""
    first = sum(2 if (X_ABS >> i) & 1 else 1 for i in range(62, i_from, -1))
    return first, first + sum(2 if (X_ABS >> i) & 1 else 1 for i in range(i_from, i_to - 1, -1))


def prog_miller_stream(i_from=62, i_to=0):
    """prog_miller_part(i_from, i_to) for lines that ARRIVE while the loop runs: before the first use of a chunk of STREAM_BOUNDS the
    built-in ACQ (first step, steps) waits for it -- the workgroup that runs prog_key_lines(0, publish=True) -- and copies it into L,
    LSCALE0X<n> evaluates its lines at P0, then the Miller steps that use them"""
    lo, hi = miller_line_steps(i_from, i_to)
    assert lo == 0 or lo in STREAM_BOUNDS
    st, step, have = [], lo, lo
    for i in range(i_from, i_to - 1, -1):
        nl = 2 if (X_ABS >> i) & 1 else 1
        while have < step + nl:
            b = min(x for x in STREAM_BOUNDS if x > have)
            st.append(('ACQ', have, b - have, 0))
            for s0 in range(have, b, 8):
                k = min(8, b - s0)
                assert k in (8, 4)
                st.append(('LSCALE0X%d' % k, ('L', 12 * s0), ('L', 12 * s0), 'P'))
            have = b
        if i != i_from:
            st.append(('SQR', 'F', 'F', 'F'))
        for _ in range(nl):
            st.append(('MUL_LINE', 'F', 'F', ('L', 12 * step)))
            step += 1
    assert step == hi and have == hi
    return st


CONJ_F = [('CONJ', 'F', 'F', 'F')]                 # x < 0: the Miller function of |x| is conjugated once, after the product


# One Miller loop on two workgroups (kernels.cuh k_pairing_post2): the accumulator after all iterations is (f_hi)^(2^SPLIT_AT) f_lo, f_hi the
# accumulation over iterations 62..SPLIT_AT and f_lo the one over the last SPLIT_AT iterations started from 1 -- squaring is
# multiplicative.  A squaring alone is half an iteration (1.9 of 3.8 us), so the first workgroup runs 22 iterations and 41 squarings
# (~170 us) while the second runs 41 iterations (~160 us) instead of one workgroup running 63 (~250 us); three workgroups (SPLIT3 below)
# come to ~140 us, and with more the squarings alone (63 x 2 us) are the floor.
SPLIT_AT = 41


def prog_miller_part(i_from, i_to):
    """the iterations i_from..i_to of prog_miller((0,)) on the accumulator F (no squaring in the first one: F is 1 there); their
    lines are evaluated at P0 first, in whole LSCALE0X4 / X8 pieces (a piece may reach into the neighbour's lines: harmless)"""
    lo, hi = miller_line_steps(i_from, i_to)
    st = []
    s0 = lo - lo % 4
    while s0 < hi:
        k = 8 if hi - s0 > 4 else 4
        assert s0 + k <= NSTEPS
        st.append(('LSCALE0X%d' % k, ('L', 12 * s0), ('L', 12 * s0), 'P'))
        s0 += k
    step = lo
    for i in range(i_from, i_to - 1, -1):
        if i != i_from:
            st.append(('SQR', 'F', 'F', 'F'))
        for _ in range(2 if (X_ABS >> i) & 1 else 1):
            st.append(('MUL_LINE', 'F', 'F', ('L', 12 * step)))
            step += 1
    assert step == hi
    return st


SPLIT3 = (54, 36)                # the same on three workgroups: 9 iterations and 54 squarings, 18 and 36, 36 (~140 us each)


def prog_post_part(points, part):
    """POST with its Miller loop cut at the iterations `points` (descending): part k runs iterations b[k] - 1 .. b[k + 1] of b = [63] + points
    + [0] on an accumulator started from 1 and squares b[k + 1] more times (the product of the parts is the whole loop's value).  Part 0
    then takes the other parts' values from its partners (built-in ACQF slot -> U) and goes on with the rest of POST; the others hand
    theirs over (built-in PUBF, slot = part - 1)"""
    b = [63] + list(points) + [0]
    st = prog_pprep((0,)) + prog_miller_part(b[part] - 1, b[part + 1]) + [('SQR', 'F', 'F', 'F')] * b[part + 1]
    if part:
        return st + [('PUBF', 'F', part - 1, 0)]
    for slot in range(len(points)):
        st += [('ACQF', 'U', slot, 0), ('MUL', 'F', 'F', 'U')]
    return st + [('MUL', 'F', 'F', 'W')] + CONJ_F + prog_easy() + prog_final_hard()


def prog_f12_inv():
    """T <- F^-1 with U = conj(F) given: F conj(F) lies in Fp6 (even coefficients); FPINV is the interpreter's one built-in, an
    inversion in Fp on a lone lane.  W is scratch."""
    return [('MUL', 'T', 'F', 'U'), ('F6INV1', 'T', 'T', 'T'), ('F6INV2', 'W', 'T', 'T'), ('F6INV3', 'W', 'W', 'W'),
            ('FPINV', ('W', 3), ('W', 2), ('W', 2)), ('F6INV4', 'W', 'W', 'W'), ('F6INV5', 'T', 'T', 'W'), ('MUL', 'T', 'U', 'T')]


def prog_easy():
    """F <- F^((p^6 - 1)(p^2 + 1))"""
    return [('CONJ', 'U', 'F', 'F')] + prog_f12_inv() + [('MUL', 'F', 'U', 'T'), ('FROB2', 'T', 'F', 'F'), ('MUL', 'F', 'T', 'F')]


def prog_f12_tree16():
    """L[0..11] <- the product of the sixteen Fp12 values L[12 j .. 12 j + 11], j < 16 (the line array doubles as the input store):
    the last levels of a pairing product's fold tree (kernels.cuh k_f12_tree_wide), fifteen general products"""
    st, d = [], 1
    while d < 16:
        st += [('MUL', ('L', 12 * j), ('L', 12 * j), ('L', 12 * (j + d))) for j in range(0, 16, 2 * d)]
        d *= 2
    return st


def prog_horner():
    """F <- conj of the Horner chain over the 68 Fp12 values L[12 e .. 12 e + 11]: F = P_0, then per Miller entry e >= 1 a squaring
    (doubling entries only) and F <- F P_e.  With P_e = the product over the items of a pairing product of their line values at
    entry e this is the product of the items' Miller functions (squaring is multiplicative): kernels.cuh k_f12_horner_wide"""
    st, step = [], 0
    for i in range(62, -1, -1):
        if i != 62:
            st.append(('SQR', 'F', 'F', 'F'))
        for _ in range(2 if (X_ABS >> i) & 1 else 1):
            st.append(('COPY', 'F', ('L', 0), ('L', 0)) if step == 0 else ('MUL', 'F', 'F', ('L', 12 * step)))
            step += 1
    assert step == NSTEPS
    return st + [('CONJ', 'F', 'F', 'F')]


PROGRAMS = [('FINAL_HARD', prog_final_hard()),
            ('FINAL', prog_easy() + prog_final_hard()),              # the whole final exponentiation of a Miller product (aggregate verify)
            # core_verify of Bls12381G1Impl: pair 1's G2 argument is the constant -g2, its lines come from a table
            ('PAIR_FIXED', prog_pprep((0, 1)) + prog_key_lines(0) + prog_miller() + CONJ_F + prog_easy() + prog_final_hard()),
            ('PAIR_GENERAL', prog_pprep((0, 1)) + prog_key_lines(0) + prog_key_lines(1) + prog_miller() + CONJ_F + prog_easy() + prog_final_hard()),
            # the same check cut where its inputs become known: the key's lines (needs the key), the Miller function of the
            # (signature, -g2) pair (needs the signature), and the rest (needs H(m)): F <- miller(pair 0) * W, W = the other function
            ('PRE_LINES', prog_key_lines(0)),
            ('PRE_F1', prog_pprep((1,)) + prog_miller((1,))),
            # Bls12381G2Impl: pair 1 is (-g1, signature): its lines come from the signature, then the same Miller function
            ('PRE_F1G', prog_key_lines(1) + prog_pprep((1,)) + prog_miller((1,))),
            ('POST', prog_pprep((0,)) + prog_miller((0,)) + [('MUL', 'F', 'F', 'W')] + CONJ_F + prog_easy() + prog_final_hard()),
            # PRE_LINES and POST side by side on workgroups of one launch, the lines handed over a few steps at a time and the Miller loop
            # itself on two of them (k_pairing_stream):
            # for the checks whose key (or H(m)) only exists when everything else is done -- the tail of a key sum, Bls12381G2Impl's hash
            ('PRE_LINES_S', prog_key_lines(0, publish=True)),
            ('POST_HI_S', prog_pprep((0,)) + prog_miller_stream(62, SPLIT_S) + [('SQR', 'F', 'F', 'F')] * SPLIT_S
             + [('ACQF', 'U', 0, 0), ('MUL', 'F', 'F', 'U'), ('MUL', 'F', 'F', 'W')] + CONJ_F + prog_easy() + prog_final_hard()),
            ('POST_LO_S', prog_pprep((0,)) + prog_miller_stream(SPLIT_S - 1, 0) + [('PUBF', 'F', 0, 0)]),
            # POST with its Miller loop on two workgroups (k_pairing_post2)
            ('POST_HI', prog_post_part((SPLIT_AT,), 0)),
            ('POST_LO', prog_post_part((SPLIT_AT,), 1)),
            # ... and on three (the default while a batch leaves every item three CUs)
            ('POST3_HI', prog_post_part(SPLIT3, 0)),
            ('POST3_MID', prog_post_part(SPLIT3, 1)),
            ('POST3_LO', prog_post_part(SPLIT3, 2)),
            ('F12_TREE16', prog_f12_tree16()),
            ('HORNER', prog_horner())]


def layout_f12():
    return Layout([('F', 12), ('T', 12), ('U', 12), ('W', 12), ('ACC', 12), ('TMP', 2 * (max(o.ntmp for o in OPS) + 1)), ('CONST', 24), ('P', 8),
                   ('PT0', 32), ('PT1', 32), ('L', 12 * NSTEPS)])


def layout_pt():
    return Layout([('L0', 8 * G2S), ('L1', 4 * G2S), ('L2', 2 * G2S), ('L3', G2S), ('L4', G2S), ('TMP', 2 * (max(o.ntmp for o in OPS_PT) + 1)),
                   ('R0', G2S), ('R1', G2S), ('R2', G2S), ('R3', G2S), ('R4', G2S), ('MS', 12), ('CONST', 64), ('ISO', 2 * ISO_STRIDE)])


# ------------------------------------------------------------------ emission
def emit_set(out, ops, lay, programs, prefix, vprefix, trait, has_inv):
    """one table set in the interpreter's format (csrc/wide_engine.cuh): byte offsets into the value store, pre-resolved"""
    tmp_base = lay.base['TMP']
    prods, lin_words, oprows = [], [], []
    for op in ops:
        nsub = (len(op.prods) + ROWS - 1) // ROWS
        bslots = {i >> 12 for _, b, _ in op.prods for _, i in b}
        aslots = {i >> 12 for a, _, _ in op.prods for _, i in a}
        assert aslots <= {SA} and len(bslots) <= 1 and bslots <= {SA, SB, CONST}, (op.name, aslots, bslots)
        bsel = 1 if bslots == {CONST} else (2 if bslots == {SA} else 0)
        maxt = max(sum(1 for _, i in t if (i >> 12) == TMP) for t, _ in op.lins)
        stride = (3 + maxt + 3) // 4                       # 16-byte chunks per linear row: header, two plain-value slots, product terms
        assert stride <= 7
        assert len(prods) < 65536 and len(lin_words) // 4 < 65536 and len(op.lins) < 256 and nsub < 256
        oprows.append((op.name, len(prods), len(lin_words) // 4, nsub, len(op.lins), stride, bsel, len(op.prods)))
        idle = (tmp_base + 2 * op.ntmp) * 64               # an idle row multiplies zero by zero into a scratch entry nothing reads
        assert idle < 65536
        for r in list(op.prods) + [None] * (nsub * ROWS - len(op.prods)):
            if r is None:
                prods.append((0, 0, 0, idle))
                continue
            a, b, o = r
            assert (o >> 12) == TMP
            a = a + [(0, 0)] * (2 - len(a))
            b = b + [(0, 0)] * (2 - len(b))
            for k, _ in a + b:
                assert -128 <= k < 128
            offs = [(i & 0xfff) * 64 for _, i in a + b]
            assert max(offs) < 65536
            cw = sum((k & 0xff) << (8 * j) for j, (k, _) in enumerate(a + b))
            prods.append((offs[0] | offs[1] << 16, offs[2] | offs[3] << 16, cw, (tmp_base + 2 * (o & 0xfff)) * 64))
        for terms, o in op.lins:
            assert (o >> 12) == DST and (o & 0xfff) * 64 < 65536
            rel = [(k, i) for k, i in terms if (i >> 12) == SA]
            ab = [(k, i) for k, i in terms if (i >> 12) == TMP]
            assert len(rel) + len(ab) == len(terms) and len(rel) <= 2, op.name
            row = [len(ab) | ((o & 0xfff) * 64) << 16]
            for k, i in rel + [(0, idx(SA, 0))] * (2 - len(rel)) + ab:
                assert -32768 <= k < 32768
                off = ((i & 0xfff) if (i >> 12) == SA else tmp_base + 2 * (i & 0xfff)) * 64
                assert off < 65536
                row.append(off | (k & 0xffff) << 16)
            row += [0] * (4 * stride - len(row))
            lin_words += row
    out.append('BLS_CONST wide_prod %s_PROD[%d] = {' % (prefix, len(prods)))
    out += ['    {0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu},' % p for p in prods]
    out.append('};')
    out.append('BLS_CONST uint32_t %s_LIN[%d] = {' % (prefix, len(lin_words)))
    for i in range(0, len(lin_words), 8):
        out.append('    ' + ', '.join('0x%08xu' % w for w in lin_words[i:i + 8]) + ',')
    out.append('};')
    out.append('BLS_CONST wide_op %s_OPS[%d] = {' % (prefix, len(oprows)))
    for name, po, lo, ns, nl, stv, bs, np_ in oprows:
        out.append('    {%d, %d, %d, %d, %d, %d},   // WOP_%s: %d products' % (po, lo, ns, nl, stv, bs, name, np_))
    out.append('};')
    names = [o[0] for o in oprows]
    for k, name in enumerate(names):
        out.append('#define WOP_%s %d' % (name, k))
    if has_inv:
        out.append('#define WOP_FPINV %d   // interpreter built-in: value dst <- (value a)^-1 in Fp on a lone lane (fp_inv_var)' % len(names))
        names.append('FPINV')
        # built-ins of the streamed cut, handled by the kernel's hook (wide_engine.cuh wide_exec): the step's dst field is the first line
        # step of a chunk, its a field the number of steps
        out.append('#define WOP_ACQ %d     // wait for line steps [dst, dst + a) of the partner workgroup and copy them into L' % len(names))
        names.append('ACQ')
        out.append('#define WOP_PUB %d     // hand line steps [dst, dst + a) of L to the partner workgroup' % len(names))
        names.append('PUB')
        out.append('#define WOP_ACQF %d    // wait for the Fp12 value in hand-over slot a of a partner workgroup and copy it into array dst' % len(names))
        names.append('ACQF')
        out.append('#define WOP_PUBF %d    // hand the Fp12 value in array dst to the partner workgroup through hand-over slot a' % len(names))
        names.append('PUBF')
    out.append('// value store (indices of 16-word values)')
    for k, v in lay.base.items():
        out.append('#define %s_%s %d' % (vprefix, k, v))
    out.append('#define %s_COUNT %d' % (vprefix, lay.count))
    out.append('// programs: two words per step: op | dst << 16,  a | b << 16  (value-store indices of the arrays)')
    for pname, st in programs:
        words = []
        for n, d, x, y in st:
            words.append('0x%08xu' % (names.index(n) | lay.ref(d) << 16))
            words.append('0x%08xu' % (lay.ref(x) | lay.ref(y) << 16))
        out.append('#define WIDE_PROG_%s_LEN %d' % (pname, len(st)))
        out.append('BLS_CONST uint32_t WIDE_PROG_%s[%d] = {' % (pname, len(words)))
        for i in range(0, len(words), 8):
            out.append('    ' + ', '.join(words[i:i + 8]) + ',')
        out.append('};')
    pmax = max(len(st) for _, st in programs)
    out.append('struct %s {' % trait)
    out.append('  enum { NPROD = %d, NLIN = %d, NOPS = %d, NV = %d, PROG_MAX = %d, CONST_BASE = %d, OP_INV = %d, FROB_CONSTS = %d };' % (
        len(prods), len(lin_words), len(oprows), lay.count, pmax, lay.base.get('CONST', -1), len(oprows) if has_inv else -1, 1 if has_inv else 0))
    out.append('  WIDE_TB_FN const wide_prod* prods() { return %s_PROD; }' % prefix)
    out.append('  WIDE_TB_FN const uint32_t* lins() { return %s_LIN; }' % prefix)
    out.append('  WIDE_TB_FN const wide_op* ops() { return %s_OPS; }' % prefix)
    out.append('};')
    return pmax


def emit(path):
    lay, lay_pt = layout_f12(), layout_pt()
    out = []
    out.append('// GENERATED by tools/gen_wide_tables.py -- do not edit.  Operation tables and programs of the row-wide engine')
    out.append('// (wide_engine.cuh).  tests/test_wide_tables.py checks every table and program on integers (Fp12 arithmetic, whole pairing checks, point sums).')
    out.append('#pragma once')
    out.append('#include "wide_rows.cuh"')
    out.append('// product row: scratch[out] = (ca0 V[A + a0] + ca1 V[A + a1]) * (cb0 V[B + b0] + cb1 V[B + b1]), UNREDUCED (28 columns, two words per')
    out.append('// lane: a scratch entry is 128 bytes).  a, b: two 16-bit BYTE offsets from the step\'s operand arrays A, B;  c: the four coefficients')
    out.append('// (int8);  out: absolute byte offset of the scratch entry')
    out.append('struct wide_prod { uint32_t a, b, c, out; };')
    out.append('// linear rows live in a word pool, `stride` 16-byte chunks per row of an operation: header nT | (byte offset of the output from the')
    out.append('// destination array) << 16, two plain-value terms (byte offset from operand array A | coefficient << 16; coefficient 0 when unused),')
    out.append('// then nT product terms (absolute byte offset of the scratch entry | coefficient << 16);  V[DST + out] = reduce(sum c_i term_i): ONE')
    out.append('// Montgomery reduction per row, after the sum (none for a row without product terms)')
    out.append('// operation: prod_off in rows, lin_off in 16-byte chunks; bsel: 0 = B from the step, 1 = the constants, 2 = B is A')
    out.append('struct wide_op { uint16_t prod_off, lin_off; uint8_t nsub, nlin, stride, bsel; };')
    out.append('#if defined(__HIPCC__)')
    out.append('#define WIDE_TB_FN static __device__ __forceinline__')
    out.append('#else')
    out.append('#define WIDE_TB_FN static inline')
    out.append('#endif')
    out.append('// ---- table set F12: Fp12 arithmetic and the Miller loop (pairing programs)')
    pmax = emit_set(out, OPS, lay, PROGRAMS, 'WIDE_F12', 'WV', 'wide_tb_f12', True)
    out.append('#define WIDE_STEPS %d' % NSTEPS)
    out.append('#define WIDE_PROG_MAX %d' % pmax)
    out.append('// ---- table set PT: sums of sixteen points per workgroup (complete projective additions)')
    emit_set(out, OPS_PT, lay_pt, PROGRAMS_PT, 'WIDE_PT', 'WPV', 'wide_tb_pt', False)
    out.append('#define WIDE_PT_ISO_STRIDE %d   // values per mapped point in array ISO (x\' = xn / xd at 0, 1; y at 2)' % ISO_STRIDE)
    out.append('#define WIDE_PT_ISO_K %d        // first isogeny coefficient in array CONST' % ISO_K)
    out.append('#define WIDE_PT_SLOT_G1 %d' % G1S)
    out.append('#define WIDE_PT_SLOT_G2 %d' % G2S)
    open(path, 'w').write('\n'.join(out) + '\n')
    open(os.path.join(os.path.dirname(path), 'wide_rows.cuh'), 'w').write(
        '// GENERATED by tools/gen_wide_tables.py -- do not edit.\n#pragma once\n'
        '#define WIDE_TABLE_ROWS %d   // DPP rows per product sub-round the engine tables are laid out for (workgroup = 16 x this many threads)\n'
        '#define WIDE_PT_POINTS %d    // points one workgroup sums with the point-sum programs (table set PT)\n' % (ROWS, PT_POINTS))


if __name__ == '__main__':
    emit(os.path.join(ROOT, 'agora-blsful_amd', 'csrc', 'wide_tables.cuh'))
    for o in OPS + OPS_PT:
        print(o.name, len(o.prods), 'products,', len(o.lins), 'linear rows, longest', max((len(t) for t, _ in o.lins), default=0))
    print('written; check with: python -m pytest tests/test_wide_tables.py')
