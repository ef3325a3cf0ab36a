"""BLS12-381 arithmetic -- pure-Python big-int ORACLE (test infrastructure, never shipped).

This file is the independent CPU restatement of the arithmetic that the reference crate
(dashpay/agora-blsful, `blsful` 3.0.0-pre8) delegates to the un-vendored `blstrs_plus 0.8`
-> `blst` dependency (reference Cargo.toml:21,23,28; re-exported at src/impls.rs:185-215).
Because that dependency is absent from /root/reference, the algorithms below restate the
*published* specifications:

  * curve / tower / optimal-ate pairing: the BLS12-381 parameterisation
    (x = -0xd201000000010000), draft-irtf-cfrg-pairing-friendly-curves;
  * hash-to-curve: RFC 9380 suites BLS12381G1_XMD:SHA-256_SSWU_RO_ and
    BLS12381G2_XMD:SHA-256_SSWU_RO_ (reference call sites src/impls/g1.rs:18, g2.rs:16);
  * point encoding: the ZCash BLS12-381 compressed serialisation
    (reference call sites src/impls/legacy.rs:88,107,132,151, src/public_key.rs:71).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity pins: see oracle/py/check_kats.py (reference KATs K1-K4 of SURVEY.md section 8c).

Representation: Fp = int mod P; Fp2 = (c0, c1) with u^2 = -1;
Fp12 = 6-tuple of Fp2 coefficients over the basis w^0..w^5 with w^6 = xi = 1 + u.
Affine points are (x, y) tuples or None for the point at infinity.
"""
import hashlib

# ---------------------------------------------------------------- parameters
X_ABS = 0xd201000000010000          # |x|; the BLS parameter is x = -X_ABS
X = -X_ABS
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
# structural self-check of the recalled constants against the BLS12 family polynomials
assert R == X**4 - X**2 + 1
assert P == ((X - 1)**2 * R) // 3 + X and ((X - 1)**2 * R) % 3 == 0
H1 = (X - 1)**2 // 3                # G1 cofactor 0x396c8c005555e1568c00aaab0000aaab
assert H1 == 0x396c8c005555e1568c00aaab0000aaab
H_EFF_G1 = 1 - X                    # 0xd201000000010001 (RFC 9380 8.8.1)

G1_GEN = (
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
)
G2_GEN = (
    (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
     0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
    (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
     0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
)


# ---------------------------------------------------------------- Fp
def fp_inv(a):
    return pow(a, -1, P)


def fp_sqrt(a):
    """Square root in Fp (P = 3 mod 4) or None."""
    a %= P
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None


def fp_is_square(a):
    a %= P
    return a == 0 or pow(a, (P - 1) // 2, P) == 1


# ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)


def f2(a, b=0):
    return (a % P, b % P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_muls(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_conj(a):
    return (a[0], -a[1] % P)


def f2_inv(a):
    n = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_mul_xi(a):
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


def f2_is_square(a):
    return fp_is_square(a[0] * a[0] + a[1] * a[1])


def f2_sqrt(a):
    """Some square root in Fp2 or None (complex method; sign is fixed by callers)."""
    if a == F2_ZERO:
        return F2_ZERO
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt(-a0 % P)           # sqrt(-1) = u
        return (0, s)
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    half = fp_inv(2)
    t = (a0 + n) * half % P
    s = fp_sqrt(t)
    if s is None:
        t = (a0 - n) * half % P
        s = fp_sqrt(t)
        if s is None:
            return None
    r = (s, a1 * fp_inv(2 * s % P) % P)
    return r if f2_sqr(r) == (a0 % P, a1 % P) else None


# ---------------------------------------------------------------- Fp12 = Fp2[w]/(w^6 - xi)
F12_ONE = (F2_ONE,) + (F2_ZERO,) * 5


def f12_mul(a, b):
    acc = [[0, 0] for _ in range(11)]
    for i in range(6):
        ai0, ai1 = a[i]
        if ai0 == 0 and ai1 == 0:
            continue
        for j in range(6):
            bj0, bj1 = b[j]
            t = acc[i + j]
            t[0] += ai0 * bj0 - ai1 * bj1
            t[1] += ai0 * bj1 + ai1 * bj0
    out = []
    for k in range(6):
        c0, c1 = acc[k]
        if k < 5:
            h0, h1 = acc[k + 6]
            c0 += h0 - h1              # * xi = (1 + u)
            c1 += h0 + h1
        out.append((c0 % P, c1 % P))
    return tuple(out)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    """a^(p^6): w -> -w."""
    return (a[0], f2_neg(a[1]), a[2], f2_neg(a[3]), a[4], f2_neg(a[5]))


# Frobenius coefficients gamma[j][k] = xi^(k (p^j - 1)/6)
def _frob_table():
    tabs = []
    for j in (1, 2, 3):
        e = (P**j - 1) // 6
        g = f2_pow(XI, e)
        row, cur = [], F2_ONE
        for _ in range(6):
            row.append(cur)
            cur = f2_mul(cur, g)
        tabs.append(row)
    return tabs


_FROB = _frob_table()


def f12_frob(a, j=1):
    """a^(p^j), j in 1..3."""
    row = _FROB[j - 1]
    if j % 2:
        return tuple(f2_mul(f2_conj(a[k]), row[k]) for k in range(6))
    return tuple(f2_mul(a[k], row[k]) for k in range(6))


def _f6_mul(a, b):
    """Fp6 = Fp2[v]/(v^3 - xi) on 3-tuples."""
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)


def _f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))


def f12_inv(a):
    """1/a = conj(a) / (a * conj(a)); a*conj(a) has only even powers of w (an Fp6 element in v = w^2)."""
    c = f12_conj(a)
    n = f12_mul(a, c)
    assert n[1] == F2_ZERO and n[3] == F2_ZERO and n[5] == F2_ZERO
    ni = _f6_inv((n[0], n[2], n[4]))
    return f12_mul(c, (ni[0], F2_ZERO, ni[1], F2_ZERO, ni[2], F2_ZERO))


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == '1':
            r = f12_mul(r, a)
    return r


# ---------------------------------------------------------------- curves (affine, None = infinity)
class Curve:
    """y^2 = x^3 + a x + b over Fp (k=1) or Fp2 (k=2); affine arithmetic."""

    def __init__(self, k, a, b):
        self.k = k
        if k == 1:
            self.add_, self.sub_, self.mul_, self.sqr_ = (
                lambda x, y: (x + y) % P, lambda x, y: (x - y) % P,
                lambda x, y: x * y % P, lambda x: x * x % P)
            self.inv_, self.neg_, self.muls_ = fp_inv, (lambda x: -x % P), (lambda x, s: x * s % P)
            self.zero = 0
        else:
            self.add_, self.sub_, self.mul_, self.sqr_ = f2_add, f2_sub, f2_mul, f2_sqr
            self.inv_, self.neg_, self.muls_ = f2_inv, f2_neg, f2_muls
            self.zero = F2_ZERO
        self.a, self.b = a, b

    def rhs(self, x):
        return self.add_(self.add_(self.mul_(self.sqr_(x), x), self.mul_(self.a, x)), self.b)

    def on_curve(self, pt):
        return pt is None or self.sqr_(pt[1]) == self.rhs(pt[0])

    def neg(self, pt):
        return None if pt is None else (pt[0], self.neg_(pt[1]))

    def dbl(self, pt):
        if pt is None or pt[1] == self.zero:
            return None
        x, y = pt
        lam = self.mul_(self.add_(self.muls_(self.sqr_(x), 3), self.a), self.inv_(self.muls_(y, 2)))
        x3 = self.sub_(self.sqr_(lam), self.muls_(x, 2))
        return (x3, self.sub_(self.mul_(lam, self.sub_(x, x3)), y))

    def add(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        if p1[0] == p2[0]:
            return self.dbl(p1) if p1[1] == p2[1] else None
        lam = self.mul_(self.sub_(p2[1], p1[1]), self.inv_(self.sub_(p2[0], p1[0])))
        x3 = self.sub_(self.sub_(self.sqr_(lam), p1[0]), p2[0])
        return (x3, self.sub_(self.mul_(lam, self.sub_(p1[0], x3)), p1[1]))

    def mul(self, pt, n):
        if n < 0:
            return self.mul(self.neg(pt), -n)
        acc = None
        for bit in bin(n)[2:] if n else '':
            acc = self.dbl(acc)
            if bit == '1':
                acc = self.add(acc, pt)
        return acc


E1 = Curve(1, 0, 4)
E2 = Curve(2, F2_ZERO, (4, 4))          # M-twist y^2 = x^3 + 4(1+u)
assert E1.on_curve(G1_GEN) and E2.on_curve(G2_GEN)


# psi: the untwist-Frobenius-twist endomorphism on E2 (used for G2 cofactor clearing and subgroup test)
_PSI_CX = f2_inv(f2_pow(XI, (P - 1) // 3))
_PSI_CY = f2_inv(f2_pow(XI, (P - 1) // 2))


def g2_psi(pt):
    if pt is None:
        return None
    return (f2_mul(f2_conj(pt[0]), _PSI_CX), f2_mul(f2_conj(pt[1]), _PSI_CY))


def g1_in_subgroup(pt):
    return E1.mul(pt, R) is None


def g2_in_subgroup(pt):
    return E2.mul(pt, R) is None


def g2_clear_cofactor(pt):
    """RFC 9380 Appendix G.3 (Budroni-Pintore): h_eff * P via psi."""
    t1 = E2.mul(pt, X)                       # c1 * P
    t2 = g2_psi(pt)
    t3 = g2_psi(g2_psi(E2.dbl(pt)))          # psi^2(2P)
    t3 = E2.add(t3, E2.neg(t2))
    t2 = E2.add(t1, t2)
    t2 = E2.mul(t2, X)
    t3 = E2.add(t3, t2)
    t3 = E2.add(t3, E2.neg(t1))
    return E2.add(t3, E2.neg(pt))


# ---------------------------------------------------------------- pairing
def _line_sparse(c0, c2, c3):
    return (c0, F2_ZERO, c2, c3, F2_ZERO, F2_ZERO)


def miller_loop(pairs):
    """prod_i f_{|x|,Q_i}(P_i), conjugated (x < 0).  pairs: [(P in E1 affine, Q in E2 affine)].

    Affine twist arithmetic.  With the untwist (x', y') -> (x'/w^2, y'/w^3) the line through T with
    slope lambda' (in Fp2), evaluated at P and scaled by w^3 (an Fp4 element, killed by the final
    exponentiation), is  (lambda' x_T - y_T) - lambda' x_P w^2 + y_P w^3.
    Pairs with an infinity member contribute 1 (as blst's multi_miller_loop does).
    """
    pairs = [(p, q) for (p, q) in pairs if p is not None and q is not None]
    f = F12_ONE
    ts = [q for (_, q) in pairs]
    bits = bin(X_ABS)[3:]
    for bit in bits:
        f = f12_sqr(f)
        for idx, (p, q) in enumerate(pairs):
            t = ts[idx]
            lam = f2_mul(f2_muls(f2_sqr(t[0]), 3), f2_inv(f2_muls(t[1], 2)))
            f = f12_mul(f, _line_sparse(f2_sub(f2_mul(lam, t[0]), t[1]),
                                        f2_neg(f2_muls(lam, p[0])), (p[1], 0)))
            x3 = f2_sub(f2_sqr(lam), f2_muls(t[0], 2))
            ts[idx] = (x3, f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1]))
        if bit == '1':
            for idx, (p, q) in enumerate(pairs):
                t = ts[idx]
                lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
                f = f12_mul(f, _line_sparse(f2_sub(f2_mul(lam, t[0]), t[1]),
                                            f2_neg(f2_muls(lam, p[0])), (p[1], 0)))
                x3 = f2_sub(f2_sub(f2_sqr(lam), t[0]), q[0])
                ts[idx] = (x3, f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1]))
    return f12_conj(f)


def _pow_x(a):
    """a^x for a in the cyclotomic subgroup (inverse = conjugate)."""
    return f12_conj(f12_pow(a, X_ABS))


# 3 (p^4 - p^2 + 1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3   (Hayashida-Hayasaka-Teruya)
assert 3 * ((P**4 - P**2 + 1) // R) == (X - 1)**2 * (X + P) * (X**2 + P**2 - 1) + 3


def final_exponentiation(f):
    """f^(3 (p^12-1)/r): cube of the canonical value; equal to 1 iff the canonical value is."""
    f = f12_mul(f12_conj(f), f12_inv(f))          # ^(p^6 - 1)
    f = f12_mul(f12_frob(f, 2), f)                # ^(p^2 + 1)
    t = f12_mul(_pow_x(f), f12_conj(f))           # f^(x-1)
    t = f12_mul(_pow_x(t), f12_conj(t))           # f^((x-1)^2)
    t = f12_mul(_pow_x(t), f12_frob(t, 1))        # ^(x+p)
    t = f12_mul(f12_mul(_pow_x(_pow_x(t)), f12_frob(t, 2)), f12_conj(t))   # ^(x^2+p^2-1)
    return f12_mul(t, f12_mul(f12_sqr(f), f))


def final_exponentiation_naive(f):
    """Generic f^((p^12-1)/r) (canonical value), for cross-checks only."""
    return f12_pow(f, (P**12 - 1) // R)


def pairing_product_is_one(pairs):
    return final_exponentiation(miller_loop(pairs)) == F12_ONE


# ---------------------------------------------------------------- ZCash compressed encoding
def _fp_lex_largest(y):
    return y > (P - 1) // 2


def _f2_lex_largest(y):
    return _fp_lex_largest(y[1]) if y[1] != 0 else _fp_lex_largest(y[0])


def g1_compress(pt):
    if pt is None:
        return bytes([0xc0]) + bytes(47)
    b = bytearray(pt[0].to_bytes(48, 'big'))
    b[0] |= 0x80 | (0x20 if _fp_lex_largest(pt[1]) else 0)
    return bytes(b)


def g2_compress(pt):
    if pt is None:
        return bytes([0xc0]) + bytes(95)
    b = bytearray(pt[0][1].to_bytes(48, 'big') + pt[0][0].to_bytes(48, 'big'))
    b[0] |= 0x80 | (0x20 if _f2_lex_largest(pt[1]) else 0)
    return bytes(b)


class DecodeError(ValueError):
    pass


def g1_decompress(b, subgroup_check=True):
    if len(b) != 48:
        raise DecodeError('length')
    c, inf, s = b[0] >> 7 & 1, b[0] >> 6 & 1, b[0] >> 5 & 1
    if not c:
        raise DecodeError('compression flag')
    x = int.from_bytes(bytes([b[0] & 0x1f]) + b[1:], 'big')
    if inf:
        if s or x:
            raise DecodeError('non-canonical infinity')
        return None
    if x >= P:
        raise DecodeError('x >= p')
    y = fp_sqrt(E1.rhs(x))
    if y is None:
        raise DecodeError('not on curve')
    if _fp_lex_largest(y) != bool(s):
        y = -y % P
    pt = (x, y)
    if subgroup_check and not g1_in_subgroup(pt):
        raise DecodeError('not in subgroup')
    return pt


def g2_decompress(b, subgroup_check=True):
    if len(b) != 96:
        raise DecodeError('length')
    c, inf, s = b[0] >> 7 & 1, b[0] >> 6 & 1, b[0] >> 5 & 1
    if not c:
        raise DecodeError('compression flag')
    x1 = int.from_bytes(bytes([b[0] & 0x1f]) + b[1:48], 'big')
    x0 = int.from_bytes(b[48:], 'big')
    if inf:
        if s or x0 or x1:
            raise DecodeError('non-canonical infinity')
        return None
    if x0 >= P or x1 >= P:
        raise DecodeError('x >= p')
    x = (x0, x1)
    y = f2_sqrt(E2.rhs(x))
    if y is None:
        raise DecodeError('not on curve')
    if _f2_lex_largest(y) != bool(s):
        y = f2_neg(y)
    pt = (x, y)
    if subgroup_check and not g2_in_subgroup(pt):
        raise DecodeError('not in subgroup')
    return pt


# ---------------------------------------------------------------- RFC 9380 hash-to-curve
def expand_message_xmd(msg, dst, n):
    if len(dst) > 255:
        dst = hashlib.sha256(b'H2C-OVERSIZE-DST-' + dst).digest()
    ell = (n + 31) // 32
    assert ell <= 255
    dst_p = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, 'big') + b'\x00' + dst_p).digest()
    bi = hashlib.sha256(b0 + b'\x01' + dst_p).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_p).digest()
        out += bi
    return out[:n]


def hash_to_field(msg, dst, count, m):
    L = 64
    u = expand_message_xmd(msg, dst, count * m * L)
    out = []
    for i in range(count):
        e = [int.from_bytes(u[L * (j + i * m):L * (j + i * m + 1)], 'big') % P for j in range(m)]
        out.append(e[0] if m == 1 else tuple(e))
    return out


def _sgn0_fp(x):
    return x & 1


def _sgn0_f2(x):
    return (x[0] & 1) | ((x[0] == 0) & (x[1] & 1))


from . import iso_consts as _iso        # noqa: E402  (derived by derive_iso.py; verified in tests)

E1_ISO = Curve(1, _iso.G1_A, _iso.G1_B)
E2_ISO = Curve(2, _iso.G2_A, _iso.G2_B)


def _sswu(curve, Z, u, is_square, sqrt, sgn0, one):
    """Simplified SWU, straight-line version of RFC 9380 6.6.2 (AB != 0)."""
    c = curve
    A, B = c.a, c.b
    zu2 = c.mul_(Z, c.sqr_(u))
    tv1 = c.add_(c.sqr_(zu2), zu2)
    if tv1 == c.zero:
        x1 = c.mul_(B, c.inv_(c.mul_(Z, A)))
    else:
        x1 = c.mul_(c.mul_(c.neg_(B), c.inv_(A)), c.add_(one, c.inv_(tv1)))
    gx1 = c.rhs(x1)
    if is_square(gx1):
        x, y = x1, sqrt(gx1)
    else:
        x = c.mul_(zu2, x1)
        y = sqrt(c.rhs(x))
    assert y is not None
    if sgn0(u) != sgn0(y):
        y = c.neg_(y)
    return (x, y)


def _horner(c, coeffs, x):
    acc = coeffs[-1]
    for k in reversed(coeffs[:-1]):
        acc = c.add_(c.mul_(acc, x), k)
    return acc


def _iso_map(c, tabs, pt):
    """Rational map (x_num/x_den, y * y_num/y_den); points hitting a pole map to infinity."""
    if pt is None:
        return None
    xn, xd, yn, yd = (_horner(c, t, pt[0]) for t in tabs)
    if xd == c.zero or yd == c.zero:
        return None
    return (c.mul_(xn, c.inv_(xd)), c.mul_(pt[1], c.mul_(yn, c.inv_(yd))))


def map_to_curve_g1(u):
    q = _sswu(E1_ISO, _iso.G1_Z, u, fp_is_square, fp_sqrt, _sgn0_fp, 1)
    return _iso_map(E1_ISO, _iso.G1_ISO, q)


def map_to_curve_g2(u):
    q = _sswu(E2_ISO, _iso.G2_Z, u, f2_is_square, f2_sqrt, _sgn0_f2, F2_ONE)
    return _iso_map(E2_ISO, _iso.G2_ISO, q)


def hash_to_g1(msg, dst):
    u0, u1 = hash_to_field(msg, dst, 2, 1)
    q = E1.add(map_to_curve_g1(u0), map_to_curve_g1(u1))
    return E1.mul(q, H_EFF_G1)


def hash_to_g2(msg, dst):
    u0, u1 = hash_to_field(msg, dst, 2, 2)
    q = E2.add(map_to_curve_g2(u0), map_to_curve_g2(u1))
    return g2_clear_cofactor(q)
