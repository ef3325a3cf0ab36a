"""Derive the RFC 9380 isogeny maps E' -> E for BLS12-381 G1 (11-isogeny) and G2 (3-isogeny).

Build-time tooling shared by the oracle tables and the device constant header.  The reference crate gets these tables from the un-vendored
`blst` dependency (reference call sites src/impls/g1.rs:18, src/impls/g2.rs:16); there is no copy of
them anywhere on this machine, so they are *computed*:

  1. division polynomial psi_l of the SSWU curve E' (l = 11 over Fp, l = 3 over Fp2);
  2. its factors of degree (l-1)/2 that are kernel polynomials of rational l-isogenies;
  3. Velu/Kohel formulas -> normalised isogeny  X = N(x)/h(x)^2,  Y = y * X'(x), codomain E*;
  4. keep the kernel whose codomain has j = 0 and compose with the isomorphism
     (x, y) -> (c x, d y), c^3 = d^2 = b/b*, onto E: y^2 = x^3 + b.
  5. of the 6 (c, d) choices pick the one matching the leading digits of RFC 9380 E.2/E.3's
     k_(1,11)/k_(3,15) [G1] and k_(1,3)/k_(3,3) [G2]; final confirmation = RFC 9380 J.9.1 / J.10.1
     hash_to_curve vectors and, for G2, the reference's C++ known-answer signatures (check_kats.py).

Run:  python tools/derive_iso.py > oracle/py/iso_consts.py   (about 10 s); tools/gen_consts.py reads the same file's
values through tools/iso_tables.py (an identical generated copy kept outside oracle/).
"""
import random
import sys

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


class Fp:
    q = P
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    neg = staticmethod(lambda a: -a % P)
    inv = staticmethod(lambda a: pow(a, -1, P))
    small = staticmethod(lambda n: n % P)
    rand = staticmethod(lambda: random.randrange(P))


class Fp2:
    q = P * P
    zero, one = (0, 0), (1, 0)
    add = staticmethod(lambda a, b: ((a[0] + b[0]) % P, (a[1] + b[1]) % P))
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P))
    mul = staticmethod(lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P))
    neg = staticmethod(lambda a: (-a[0] % P, -a[1] % P))
    small = staticmethod(lambda n: (n % P, 0))
    rand = staticmethod(lambda: (random.randrange(P), random.randrange(P)))

    @staticmethod
    def inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return (a[0] * n % P, -a[1] * n % P)


# ------------------------------------------------------------ dense polynomials, low -> high
def ptrim(F, a):
    while a and a[-1] == F.zero:
        a = a[:-1]
    return a


def padd(F, a, b):
    n = max(len(a), len(b))
    a = a + [F.zero] * (n - len(a))
    b = b + [F.zero] * (n - len(b))
    return ptrim(F, [F.add(x, y) for x, y in zip(a, b)])


def psub(F, a, b):
    return padd(F, a, [F.neg(x) for x in b])


def pmul(F, a, b):
    if not a or not b:
        return []
    out = [F.zero] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x == F.zero:
            continue
        for j, y in enumerate(b):
            out[i + j] = F.add(out[i + j], F.mul(x, y))
    return ptrim(F, out)


def pscale(F, a, k):
    return ptrim(F, [F.mul(x, k) for x in a])


def pdivmod(F, a, b):
    a = list(a)
    b = ptrim(F, b)
    nb = len(b)
    if len(a) < nb:
        return [], ptrim(F, a)
    binv = F.inv(b[-1])
    q = [F.zero] * (len(a) - nb + 1)
    for sh in range(len(a) - nb, -1, -1):
        k = F.mul(a[sh + nb - 1], binv)
        q[sh] = k
        if k != F.zero:
            for i, y in enumerate(b):
                a[sh + i] = F.sub(a[sh + i], F.mul(k, y))
    return ptrim(F, q), ptrim(F, a[:nb - 1])


def pmod(F, a, m):
    return pdivmod(F, a, m)[1]


def pmonic(F, a):
    return pscale(F, a, F.inv(a[-1]))


def pgcd(F, a, b):
    a, b = ptrim(F, a), ptrim(F, b)
    while b:
        a, b = b, pmod(F, a, b)
    return pmonic(F, a) if a else a


def ppowmod(F, base, e, m):
    r = [F.one]
    base = pmod(F, base, m)
    for bit in bin(e)[2:]:
        r = pmod(F, pmul(F, r, r), m)
        if bit == '1':
            r = pmod(F, pmul(F, r, base), m)
    return r


def pcompose_mod(F, f, g, m):
    """f(g(x)) mod m by Horner."""
    acc = []
    for c in reversed(f):
        acc = padd(F, pmod(F, pmul(F, acc, g), m), [c])
    return acc


def pderiv(F, a):
    return ptrim(F, [F.mul(F.small(i), a[i]) for i in range(1, len(a))])


def peval(F, a, x):
    acc = F.zero
    for c in reversed(a):
        acc = F.add(F.mul(acc, x), c)
    return acc


# ------------------------------------------------------------ division polynomials
def division_poly(F, A, B, ell):
    """g_n with psi_n = g_n (n odd) or y*g_n (n even); returns g_ell for odd ell."""
    s = F.small
    Fx = [B, A, F.zero, F.one]                     # x^3 + A x + B
    F2x = pmul(F, Fx, Fx)
    A2, A3, B2 = F.mul(A, A), F.mul(F.mul(A, A), A), F.mul(B, B)
    g = {0: [], 1: [F.one], 2: [s(2)],
         3: [F.neg(A2), F.mul(s(12), B), F.mul(s(6), A), F.zero, s(3)],
         4: pscale(F, [F.sub(F.neg(F.mul(s(8), B2)), A3), F.neg(F.mul(s(4), F.mul(A, B))),
                       F.neg(F.mul(s(5), A2)), F.mul(s(20), B), F.mul(s(5), A), F.zero, F.one], s(4))}
    half = F.inv(s(2))

    def get(n):
        if n in g:
            return g[n]
        m = n // 2
        if n % 2:
            a = pmul(F, get(m + 2), pmul(F, get(m), pmul(F, get(m), get(m))))
            b = pmul(F, get(m - 1), pmul(F, get(m + 1), pmul(F, get(m + 1), get(m + 1))))
            if m % 2 == 0:
                a = pmul(F, F2x, a)
            else:
                b = pmul(F, F2x, b)
            g[n] = psub(F, a, b)
        else:
            a = pmul(F, get(m + 2), pmul(F, get(m - 1), get(m - 1)))
            b = pmul(F, get(m - 2), pmul(F, get(m + 1), get(m + 1)))
            g[n] = pscale(F, pmul(F, get(m), psub(F, a, b)), half)
        return g[n]

    return get(ell)


def equal_degree_split(F, f, d):
    """Cantor-Zassenhaus: all monic irreducible degree-d factors of squarefree f (all of degree d)."""
    f = pmonic(F, f)
    if len(f) - 1 == d:
        return [f]
    while True:
        r = [F.rand() for _ in range(len(f) - 1)]
        t = ppowmod(F, r, (F.q**d - 1) // 2, f)
        g = pgcd(F, psub(F, t, [F.one]), f)
        if 0 < len(g) - 1 < len(f) - 1:
            return equal_degree_split(F, g, d) + equal_degree_split(F, pdivmod(F, f, g)[0], d)


def kernel_candidates(F, psi, ell):
    """Monic factors of psi of degree (ell-1)/2 that can be kernel polynomials (irreducible factors of
    degree dividing (ell-1)/2, grouped)."""
    n = (ell - 1) // 2
    psi = pmonic(F, psi)
    xq = ppowmod(F, [F.zero, F.one], F.q, psi)             # x^q mod psi
    out = []
    lin = pgcd(F, psub(F, xq, [F.zero, F.one]), psi)
    print(f'#   degree of rational-root part: {len(lin) - 1}', file=sys.stderr)
    roots = []
    if len(lin) > 1:
        roots = [F.neg(f[0]) for f in equal_degree_split(F, lin, 1)]
    if n == 1:
        return [[F.neg(r), F.one] for r in roots]
    # degree-n irreducible factors: x^(q^n) == x
    cur = xq
    for _ in range(n - 1):
        cur = pcompose_mod(F, cur, xq, psi)
    big = pgcd(F, psub(F, cur, [F.zero, F.one]), psi)
    if len(lin) > 1:
        big = pdivmod(F, big, lin)[0]
    print(f'#   degree of degree-{n}-irreducible part: {len(big) - 1}', file=sys.stderr)
    if len(big) > 1:
        out += equal_degree_split(F, big, n)
    if len(roots) == n:
        out.append(lin)                 # exactly one rational kernel with rational x-coordinates
    elif roots:
        raise NotImplementedError('several kernels with rational x-coordinates: group roots by x-only multiplication')
    return out


def velu(F, A, B, h, ell):
    """Velu/Kohel normalised isogeny with kernel polynomial h (odd degree ell); h may be irreducible."""
    assert len(pmonic(F, h)) - 1 == (ell - 1) // 2
    return _velu_by_interp(F, A, B, h, ell)


def _velu_by_interp(F, A, B, h, ell):
    """Compute N(x) = X(x) h(x)^2 exactly.

    With D = h, S1 = sum 1/(x-xQ) = D'/D and S2 = sum 1/(x-xQ)^2 = (D'^2 - D D'')/D^2:
      sum_Q v_Q/(x-xQ):  v_Q = 6xQ^2 + 2A and  xQ^2/(x-xQ) = x^2/(x-xQ) - (x + xQ), so
          = (6x^2+2A) S1 - 6 (n x + p1)
      sum_Q u_Q/(x-xQ)^2: u_Q = 4 f(xQ), f = x^3+Ax+B, and
          f(xQ) = f(x) - f'(x)(x-xQ) + (f''(x)/2)(x-xQ)^2 - (x-xQ)^3      (Taylor at x, f''' = 6)
          = 4 [ f S2 - f' S1 + 3x n - (n x - p1) ]
    """
    s = F.small
    n = (ell - 1) // 2
    D = pmonic(F, h)
    D1 = pderiv(F, D)
    D2 = pderiv(F, D1)
    p1 = F.neg(D[-2])                                   # sum of roots
    f = [B, A, F.zero, F.one]
    f1 = [A, F.zero, s(3)]
    Dsq = pmul(F, D, D)
    S1num = pmul(F, D1, D)                              # S1 = S1num / D^2
    S2num = psub(F, pmul(F, D1, D1), pmul(F, D, D2))    # S2 = S2num / D^2
    term_v = psub(F, pmul(F, [F.mul(s(2), A), F.zero, s(6)], S1num),
                  pmul(F, [F.mul(s(6), p1), s(6 * n)], Dsq))
    inner = padd(F, psub(F, pmul(F, f, S2num), pmul(F, f1, S1num)),
                 pmul(F, [p1, s(2 * n)], Dsq))          # 3xn - (nx - p1) = 2n x + p1
    term_u = pscale(F, inner, s(4))
    N = padd(F, pmul(F, [F.zero, F.one], Dsq), padd(F, term_v, term_u))
    # codomain from power sums p1,p2,p3 of the roots (Newton identities on D)
    e = [F.one] + [F.zero] * 3
    for k in range(1, 4):
        if n - k >= 0:
            c = D[n - k]
            e[k] = c if k % 2 == 0 else F.neg(c)
    p1 = e[1]
    p2 = F.sub(F.mul(e[1], p1), F.mul(s(2), e[2]))
    p3 = F.add(F.sub(F.mul(e[1], p2), F.mul(e[2], p1)), F.mul(s(3), e[3]))
    v = F.add(F.mul(s(6), p2), F.mul(s(2 * n), A))
    w = F.add(F.add(F.mul(s(10), p3), F.mul(F.mul(s(6), A), p1)), F.mul(s(4 * n), B))
    A_star = F.sub(A, F.mul(s(5), v))
    B_star = F.sub(B, F.mul(s(7), w))
    # Y = y * X'(x) = y (N' D - 2 N D') / D^3
    Ynum = psub(F, pmul(F, pderiv(F, N), D), pscale(F, pmul(F, N, D1), s(2)))
    return N, Dsq, Ynum, pmul(F, Dsq, D), A_star, B_star


def roots_of(F, f):
    """All roots in F of a small polynomial f."""
    f = pmonic(F, f)
    xq = ppowmod(F, [F.zero, F.one], F.q, f)
    g = pgcd(F, psub(F, xq, [F.zero, F.one]), f)
    if len(g) <= 1:
        return []
    return [F.neg(t[0]) for t in equal_degree_split(F, g, 1)]


def derive(F, A, B, b_target, ell, want_c_prefix, want_d_prefix, fmt):
    psi = division_poly(F, A, B, ell)
    print(f'# deg psi_{ell} = {len(psi) - 1}', file=sys.stderr)
    cands = kernel_candidates(F, psi, ell)
    print(f'# {len(cands)} rational kernel candidate(s)', file=sys.stderr)
    sols = []
    for h in cands:
        N, Dsq, Yn, Yd, As, Bs = velu(F, A, B, h, ell)
        print(f'#   codomain A* = {fmt(As)[:24]}..', file=sys.stderr)
        if As != F.zero:
            continue
        ratio = F.mul(b_target, F.inv(Bs))
        cs = roots_of(F, [F.neg(ratio), F.zero, F.zero, F.one])
        ds = roots_of(F, [F.neg(ratio), F.zero, F.one])
        for c in cs:
            for d in ds:
                sols.append((pscale(F, N, c), Dsq, pscale(F, Yn, d), Yd, c, d))
    print(f'# {len(sols)} (kernel, c, d) combinations', file=sys.stderr)
    pick = [sol for sol in sols if fmt(sol[4]).startswith(want_c_prefix) and fmt(sol[5]).startswith(want_d_prefix)]
    assert len(pick) == 1, [(fmt(s_[4])[:20], fmt(s_[5])[:20]) for s_ in sols]
    return pick[0][:4]


def hx(a):
    return '%096x' % a


def hx2(a):
    return hx(a[0]) + '+' + hx(a[1])


def emit_fp(name, tabs):
    print(f'{name} = (')
    for t in tabs:
        print('    [' + ', '.join('0x%x' % c for c in t) + '],')
    print(')')


def emit_fp2(name, tabs):
    print(f'{name} = (')
    for t in tabs:
        print('    [' + ', '.join('(0x%x, 0x%x)' % c for c in t) + '],')
    print(')')


def main():
    random.seed(381)
    # RFC 9380 8.8.1: E'1: y^2 = x^3 + A' x + B', Z = 11  (recalled; validated in tests by #E'(Fp) = #E(Fp))
    A1 = 0x144698a3b8e9433d693a02c96d4982b0ea985383ee66a8d8e8981aefd881ac98936f8da0e0f97f5cf428082d584c1d
    B1 = 0x12e2908d11688030018b12e8753eee3b2016c1f0f24f4070a0b9c14fcef35ef55a23215a316ceaa5d1cc48e98e172be0
    # RFC 9380 8.8.2: E'2: A' = 240 u, B' = 1012 (1 + u), Z = -(2 + u)
    A2, B2 = (0, 240), (1012, 1012)
    print('"""GENERATED by tools/derive_iso.py -- do not edit.  RFC 9380 8.8 SSWU curves and the derived')
    print('isogeny tables (x_num, x_den, y_num, y_den), coefficients low -> high."""')
    print('G1_A = 0x%x' % A1)
    print('G1_B = 0x%x' % B1)
    print('G1_Z = 11')
    print('G2_A = (0x%x, 0x%x)' % A2)
    print('G2_B = (0x%x, 0x%x)' % B2)
    print('G2_Z = (0x%x, 0x%x)' % ((-2) % P, (-1) % P))
    g1 = derive(Fp, A1, B1, 4, 11, '06e08c248e260e70', '15e6be4e990f03ce', hx)
    emit_fp('G1_ISO', g1)
    g2 = derive(Fp2, A2, B2, (4, 4), 3, '171d6541fa38ccfaed6d', '124c9ad43b6cf79b', hx2)
    emit_fp2('G2_ISO', g2)


if __name__ == '__main__':
    main()
