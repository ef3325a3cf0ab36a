"""The device arithmetic headers compiled as host C++ (tests/hostsim) against the oracle: every per-item device function
is checked on the CPU, bit for bit, before any kernel runs on a GPU."""
import ctypes
import json
import os
import random

import pytest

import util
from util import c, ref, P


def test_fp_ops(hs):
    rng = random.Random(1)
    edge = [0, 1, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2**380, 2**381 % P]
    vals = edge + [rng.randrange(P) for _ in range(300)]
    out = ctypes.create_string_buffer(48)
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        hs.hs_fp_mul(util.fp_raw(a), util.fp_raw(b), out)
        assert util.fp_from_raw(out.raw) == a * b % P
    o = ctypes.create_string_buffer(62 * 4)
    for a in vals[:60]:
        b = rng.randrange(P)
        hs.hs_fp_ops(util.fp_raw(a), util.fp_raw(b), o)
        r = o.raw
        assert util.fp_from_raw(r[0:48]) == (a + b) % P and util.fp_from_raw(r[48:96]) == (a - b) % P
        assert util.fp_from_raw(r[96:144]) == -a % P
        assert util.fp_from_raw(r[144:192]) == (pow(a, -1, P) if a else 0)
        assert bool(int.from_bytes(r[192:196], 'little')) == c.fp_is_square(a)
        if c.fp_is_square(a):
            assert util.fp_from_raw(r[196:244]) ** 2 % P == a
        assert bool(int.from_bytes(r[244:248], 'little')) == (a > (P - 1) // 2)
    # the variable-time inversion (public operands, one lane per item: the row-wide engine's Fp12 inversion)
    for a in vals + [2 ** k for k in range(0, 381, 13)] + [P - 2 ** k for k in range(1, 380, 17)]:
        hs.hs_fp_inv_var(util.fp_raw(a), out)
        assert util.fp_from_raw(out.raw) == (pow(a, -1, P) if a else 0)


def test_fp_from_raw_without_a_multiplication(hs):
    """fp_from_raw since round 4: the raw integer shifted left by eight bits (R = 2^384 -> 2^392) and ONE value reduction, no
    multiplication.  Against the integers and against the multiplication form of rounds 1-3, on canonical words, on the edges, and
    on NON-canonical words up to 2^384 - 1 (2,522 p after the shift: the quotient estimate's largest input; the bound tracker checks
    the stated limits on the way)."""
    rng = random.Random(44)
    R384 = 1 << 384
    M = (1 << 28) - 1
    vals = [0, 1, 2, P - 1, P, P + 1, 2 * P - 1, R384 - 1, R384 - P, (R384 // P) * P, (R384 // P) * P - 1, 1 << 383, (1 << 376) - 1, 0xff << 376]
    # integers whose shifted form sits at the rounding boundary of the quotient estimate: (k + 1/2) p / 256 for many k
    vals += [((2 * k + 1) * P // 512 + d) % R384 for k in (0, 1, 7, 100, 1260, 2520, 2521) for d in (-1, 0, 1)]
    vals += [rng.randrange(R384) for _ in range(2000)] + [rng.randrange(P) for _ in range(1000)]
    out = (ctypes.c_int32 * 28)()
    for v in vals:
        words = v.to_bytes(48, 'little')
        assert hs.hs_from_raw_forms(words, out) == 1, hex(v)
        got = sum((out[i] & 0xffffffff if i < 13 else out[i]) << (28 * i) for i in range(14))
        assert got == (v << 8) % P and all(0 <= out[i] <= M for i in range(13)), hex(v)     # the element a * 2^392 for raw = a * 2^384


def test_fp_to_raw_without_a_multiplication(hs):
    """fp_to_raw since round 4: canonical integer, one radix-2^8 Montgomery digit, one conditional subtraction.  Against the integers
    (internal value v = e 2^392 -> words of e 2^384 mod p = v / 2^8 mod p) and against the multiplication form, on values whose
    quotient lands on both sides of the conditional subtraction and on redundant signed limbs."""
    rng = random.Random(45)
    inv256 = pow(256, -1, P)
    M = (1 << 28) - 1
    vals = [0, 1, 255, 256, 257, P - 1, P - 256, P - 255, (P - 1) // 2]
    vals += [(256 * t - m * P) % P for t in (P - 1, P - 2, 0, 1, P // 2) for m in (0, 1, 128, 255)]     # T + m p = 256 t: t at the top and bottom of the range
    vals += [rng.randrange(P) for _ in range(3000)]
    out = (ctypes.c_uint32 * 12)()
    for v in vals:
        limbs = [(v >> (28 * i)) & M for i in range(14)]
        if rng.random() < 0.5:                    # a redundant form of the same integer: borrow between neighbours, add a few p
            k = rng.randint(-3, 3)
            vv = v + k * P
            limbs = [((vv >> (28 * i)) & M) for i in range(13)] + [vv >> (28 * 13)]
            for i in range(13):
                d = rng.randint(-5, 5)
                limbs[i] += d << 28
                limbs[i + 1] -= d
        arr = (ctypes.c_int32 * 14)(*limbs)
        assert hs.hs_to_raw_forms(arr, ctypes.c_double(max(abs(x) for x in limbs) + 1), ctypes.c_double(5.0), out) == 1, hex(v)
        got = sum(int(out[i]) << (32 * i) for i in range(12))
        assert got == v * inv256 % P, hex(v)


def test_fp_lazy_limbs(hs):
    """fp_norm / fp_reduce / fp_canon / fp_is_zero on redundant signed-limb vectors: exact multiples of p in lazy form,
    values at the rounding boundary of the quotient estimate, negative values, maximal limb magnitudes."""
    rng = random.Random(5)
    M = (1 << 28) - 1

    def val(l):
        return sum(int(x) << (28 * i) for i, x in enumerate(l))

    def limbs_of(v):          # canonical limbs of a non-negative integer < 2^392
        return [(v >> (28 * i)) & M for i in range(14)]

    def scramble(l, spread):  # same integer, redundant limbs: move multiples of 2^28 between neighbours
        l = list(l)
        for i in range(13):
            d = rng.randint(-spread, spread)
            l[i] += d << 28
            l[i + 1] -= d
        return l

    cases = []
    for k in (-100, -7, -3, -1, 0, 1, 2, 5, 64, 100):                       # exact multiples of p
        v = k * P
        base = limbs_of(v % (1 << 392))
        if v < 0:
            base[13] -= 1 << 28                                              # two's-complement top limb -> signed value
        assert val(base) == v
        cases.append(base)
        cases.append(scramble(base, 3))
    for k in (-9, -1, 0, 1, 8):                                              # around the rounding boundary k p +- p/2
        for d in (-2, -1, 0, 1, 2):
            v = k * P + P // 2 + d
            base = limbs_of(v % (1 << 392))
            if v < 0:
                base[13] -= 1 << 28
            cases.append(scramble(base, 2))
    for _ in range(300):                                                     # random values within +-110 p, lazy limbs
        v = rng.randrange(-110 * P, 110 * P)
        base = limbs_of(v % (1 << 392))
        if v < 0:
            base[13] -= 1 << 28
        cases.append(scramble(base, rng.choice((0, 1, 3, 7))))
    out = (ctypes.c_int32 * 55)()
    for l in cases:
        v = val(l)
        assert all(abs(x) < 2**31 - 2**8 for x in l)
        lb = float(max(abs(x) for x in l) + 1)
        arr = (ctypes.c_int32 * 14)(*l)
        hs.hs_fp_lazy(arr, ctypes.c_double(lb), ctypes.c_double(abs(v) / P + 1e-9), out)
        o = list(out)
        nrm, red, can = o[0:14], o[14:28], o[28:42]
        assert val(nrm) == v and all(-16 <= x <= M + 16 for x in nrm[:13])
        assert (val(red) - v) % P == 0 and all(0 <= x <= M for x in red[:13]) and abs(val(red)) * 100 <= 52 * P
        assert val(can) == v % P and all(0 <= x <= M for x in can)
        assert o[42] == (1 if v % P == 0 else 0)
        words = [x & 0xffffffff for x in o[43:55]]
        assert sum(w << (32 * i) for i, w in enumerate(words)) == v % P


def test_fp_reduce_lin2(hs):
    """fp_reduce_lin2 (csrc/fp.cuh: the value reduction of ka a + kb b in one pass on 64 bits, the 3 t +- 2 z of the compressed
    squarings): against integers on redundant signed-limb operands -- limb magnitudes up to what three products' sums and a xi-twist
    leave (6 x 2^28), coefficient pairs the kernels use and larger ones, values at the rounding boundary of the quotient estimate."""
    rng = random.Random(15)
    M = (1 << 28) - 1

    def val(l):
        return sum(int(x) << (28 * i) for i, x in enumerate(l))

    def lazy(v, spread):      # redundant limbs of v: multiples of 2^28 moved between neighbours
        base = [(v % (1 << 392) >> (28 * i)) & M for i in range(14)]
        if v < 0:
            base[13] -= 1 << 28
        for i in range(13):
            d = rng.randint(-spread, spread)
            base[i] += d << 28
            base[i + 1] -= d
        assert val(base) == v
        return base

    out = (ctypes.c_int32 * 14)()
    cases = []
    for ka, kb in ((3, 2), (3, -2), (1, 1), (12, 0), (1, -12), (-3, 5)):
        for _ in range(60):
            va = rng.randrange(-4 * P, 4 * P)
            vb = rng.randrange(-4 * P, 4 * P)
            cases.append((ka, kb, lazy(va, rng.choice((0, 1, 5))), lazy(vb, rng.choice((0, 1, 5)))))
        for k in (-7, 0, 3):                                # ka a + kb b at k p + p/2 +- a little (kb b = 0 there)
            for d in (-1, 0, 1):
                v = k * P + P // 2 + d
                if v % ka == 0:
                    cases.append((ka, kb, lazy(v // ka, 2), lazy(0, 2)))
    for ka, kb, la, lb in cases:
        va, vb = val(la), val(lb)
        lba = float(max(abs(x) for x in la) + 1)
        lbb = float(max(abs(x) for x in lb) + 1)
        hs.hs_fp_reduce_lin2((ctypes.c_int32 * 14)(*la), ctypes.c_double(lba), ctypes.c_double(abs(va) / P + 1e-9), ka,
                             (ctypes.c_int32 * 14)(*lb), ctypes.c_double(lbb), ctypes.c_double(abs(vb) / P + 1e-9), kb, out)
        r = list(out)
        want = ka * va + kb * vb
        assert (val(r) - want) % P == 0 and all(0 <= x <= M for x in r[:13]) and abs(val(r)) * 100 <= 52 * P, (ka, kb)


def test_fp2_one_lane_karatsuba(hs):
    """fp2_kara_products / fp2_kara_diffs (csrc/fp.cuh: a whole Fp2 product on one lane, three product streams and two interleaved
    reductions; the difference form (a - b)(a - xi b) that the compressed squarings take without a carry pass) against the integers,
    edge values included, bound tracker on."""
    rng = random.Random(44)
    edge = [0, 1, P - 1, (P - 1) // 2, (P + 1) // 2, 2**380, P - 2**200]
    vals = [(x, y) for x in edge for y in edge[:4]] + [(rng.randrange(P), rng.randrange(P)) for _ in range(200)]
    out = ctypes.create_string_buffer(4 * 48)
    xi = (1, 1)
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[(i * 5 + 2) % len(vals)]
        hs.hs_fp2_kara(util.fp_raw(a[0]) + util.fp_raw(a[1]), util.fp_raw(b[0]) + util.fp_raw(b[1]), out)
        r = out.raw
        assert (util.fp_from_raw(r[0:48]), util.fp_from_raw(r[48:96])) == c.f2_mul(a, b)
        want = c.f2_mul(c.f2_sub(a, b), c.f2_sub(a, c.f2_mul(xi, b)))
        assert (util.fp_from_raw(r[96:144]), util.fp_from_raw(r[144:192])) == want


def test_fp12_ops(hs):
    rng = random.Random(2)
    for _ in range(5):
        fa = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
        fb = tuple((rng.randrange(P), rng.randrange(P)) for _ in range(6))
        out = ctypes.create_string_buffer(144 * 4 * 5)
        hs.hs_fp12_check(util.f12_raw(fa), util.f12_raw(fb), out)
        o = out.raw
        assert util.f12_from_plain_words(o[0:576]) == c.f12_mul(fa, fb)
        assert util.f12_from_plain_words(o[576:1152]) == c.f12_sqr(fa)
        assert util.f12_from_plain_words(o[1152:1728]) == c.f12_inv(fa)
        assert util.f12_from_plain_words(o[1728:2304]) == c.f12_frob(fa, 1)
        assert util.f12_from_plain_words(o[2304:2880]) == c.f12_frob(fa, 2)


def test_hash_to_curve(hs):
    msgs = [b'', b'abc', b'hello', bytes(range(200)), b'x' * 55, b'y' * 56, b'z' * 64, b'w' * 119, b'v' * 1000]
    for m in msgs:
        for C in (ref.G1Impl, ref.G2Impl):
            for dst in list(C.DST.values()) + [C.POP_DST, b'', b'D' * 255]:
                if C is ref.G1Impl:
                    out = ctypes.create_string_buffer(48)
                    hs.hs_hash_to_g1(m, len(m), dst, len(dst), out)
                    assert out.raw == c.g1_compress(c.hash_to_g1(m, dst))
                else:
                    out = ctypes.create_string_buffer(96)
                    hs.hs_hash_to_g2(m, len(m), dst, len(dst), out)
                    assert out.raw == c.g2_compress(c.hash_to_g2(m, dst))


def test_hash_to_curve_word_form_boundaries(hs):
    """expand_message_xmd in words (csrc/h2c.cuh expand_message_xmd_words, what the lane kernels run): whole-word messages at an
    aligned and at an unaligned address, tags on both sides of the sizes where the tail of a block changes shape (one or two
    blocks per b_i: 21 / 22 bytes; the cached tail words: 93 / 94 bytes), and the byte-streaming fallback for other lengths."""
    for mlen in (0, 4, 32, 36, 60, 64, 33):
        backing = bytearray(b'\x00' + bytes((7 * i + mlen) & 0xff for i in range(mlen)) + b'\x00' * 8)
        for off in (0, 1):                               # the same message at two alignments
            buf = (ctypes.c_char * (mlen + 8)).from_buffer(backing, off)
            m = bytes(backing[off:off + mlen])
            for dlen in (0, 1, 20, 21, 22, 23, 43, 54, 55, 56, 86, 87, 92, 93, 94, 95, 128, 255):
                dst = bytes((11 * i + 3) & 0xff for i in range(dlen))
                out = ctypes.create_string_buffer(48)
                hs.hs_hash_to_g1(buf, mlen, dst, dlen, out)
                assert out.raw == c.g1_compress(c.hash_to_g1(m, dst)), (mlen, off, dlen)
        out = ctypes.create_string_buffer(96)
        for dlen in (22, 43, 94):
            dst = bytes((5 * i + 1) & 0xff for i in range(dlen))
            hs.hs_hash_to_g2(bytes(backing[1:1 + mlen]), mlen, dst, dlen, out)
            assert out.raw == c.g2_compress(c.hash_to_g2(bytes(backing[1:1 + mlen]), dst)), (mlen, dlen)


def test_group_ops(hs):
    rng = random.Random(3)
    for _ in range(4):
        k = rng.randrange(c.R)
        p1, p2 = c.E2.mul(c.G2_GEN, rng.randrange(1, c.R)), c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
        om, oa = ctypes.create_string_buffer(96), ctypes.create_string_buffer(96)
        for a, b in ((p1, p2), (p1, p1), (p1, c.E2.neg(p1)), (None, p2), (p1, None)):
            hs.hs_g2_mul_add(util.g2_raw(a, rng) if a else util.g2_raw(None), util.g2_raw(b, rng) if b else util.g2_raw(None), util.scalar_raw(k), om, oa)
            assert om.raw == c.g2_compress(c.E2.mul(a, k)) and oa.raw == c.g2_compress(c.E2.add(a, b))
        g1a, g1b = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R)), c.E1.mul(c.G1_GEN, rng.randrange(1, c.R))
        om, oa = ctypes.create_string_buffer(48), ctypes.create_string_buffer(48)
        for legacy in (0, 1):
            for a, b in ((g1a, g1b), (g1a, g1a), (g1a, c.E1.neg(g1a))):
                hs.hs_g1_mul_add(util.g1_raw(a, rng), util.g1_raw(b, rng), util.scalar_raw(k), om, oa, legacy)
                wm, wa = c.g1_compress(c.E1.mul(a, k)), c.g1_compress(c.E1.add(a, b))
                if legacy:
                    wm, wa = ref.modern_to_legacy(wm), ref.modern_to_legacy(wa)
                assert om.raw == wm and oa.raw == wa
    x = (5, 7)
    while c.f2_sqrt(c.E2.rhs(x)) is None:
        x = (x[0] + 1, x[1])
    pt = (x, c.f2_sqrt(c.E2.rhs(x)))
    out = ctypes.create_string_buffer(96)
    hs.hs_g2_clear_cofactor(util.g2_raw(pt, rng), out)
    assert out.raw == c.g2_compress(c.g2_clear_cofactor(pt))


def test_g2_split_point_ops(hs):
    """Jacobian G2 arithmetic instantiated over the lane-split Fp2 (host emulation, bound tracker on) vs the oracle."""
    rng = random.Random(9)
    for _ in range(3):
        a = c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
        b = c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
        oa, od, oc = (ctypes.create_string_buffer(96) for _ in range(3))
        hs.hs_g2_split_ops(util.g2_raw(a, rng), util.g2_raw(b, rng), oa, od, oc)
        assert oa.raw == c.g2_compress(c.E2.add(a, b))
        assert od.raw == c.g2_compress(c.E2.add(a, a))
        assert oc.raw == c.g2_compress(c.g2_clear_cofactor(a))
    # equal and opposite operands take the exceptional branches
    a = c.E2.mul(c.G2_GEN, 12345)
    oa, od, oc = (ctypes.create_string_buffer(96) for _ in range(3))
    hs.hs_g2_split_ops(util.g2_raw(a, rng), util.g2_raw(a, rng), oa, od, oc)
    assert oa.raw == c.g2_compress(c.E2.add(a, a))
    hs.hs_g2_split_ops(util.g2_raw(a, rng), util.g2_raw(c.E2.neg(a), rng), oa, od, oc)
    assert oa.raw == c.g2_compress(None)


def test_pairing_values(hs):
    """GT values (after the final exponentiation) equal the oracle's exactly; cyclotomic squaring == generic squaring."""
    rng = random.Random(4)
    P1, Q1 = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R)), c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
    P2, Q2 = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R)), c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
    out, outm = ctypes.create_string_buffer(576), ctypes.create_string_buffer(576)
    hs.hs_pairing(1, util.g1_aff_raw(P1), util.g2_aff_raw(Q1), out, outm)
    want = c.final_exponentiation(c.miller_loop([(P1, Q1)]))
    assert util.f12_from_plain_words(out.raw) == want
    assert c.final_exponentiation(util.f12_from_plain_words(outm.raw)) == want      # Miller values differ by subfield factors only
    hs.hs_pairing(2, util.g1_aff_raw(P1) + util.g1_aff_raw(P2), util.g2_aff_raw(Q1) + util.g2_aff_raw(Q2), out, None)
    assert util.f12_from_plain_words(out.raw) == c.final_exponentiation(c.miller_loop([(P1, Q1), (P2, Q2)]))
    assert hs.hs_cyclotomic_check(1, util.g1_aff_raw(P1), util.g2_aff_raw(Q1)) == 1
    assert hs.hs_pow_x_compressed_check(util.g1_aff_raw(P1), util.g2_aff_raw(Q1)) == 1   # Karabina chain == plain chain
    # round 4: the compressed squarings with one lane per Fp4 squaring (one-lane Karatsuba products, k_finalexp2s fx_pow_run) against
    # the lane-split squarings and the one-lane tower, coordinate by coordinate over a whole run of 63; and the whole chain in both forms
    assert hs.hs_cyc_kara_check(util.g1_aff_raw(P1), util.g2_aff_raw(Q1), 63) == 1
    for on in (0, 1):
        hs.hs_set_cyc_kara(on)
        assert hs.hs_pow_x_compressed_check(util.g1_aff_raw(P2), util.g2_aff_raw(Q2)) == 1
    hs.hs_set_cyc_kara(1)
    assert hs.hs_pow_x_identity_check() == 1     # a = 1: the compressed chains decline (z2 = 0) and the plain chain answers
    # the precomputed -g2 line table (tools/gen_g2_lines.py) reproduces the generic loop's Miller value bit for bit
    assert hs.hs_miller_fixed_g2_matches(util.g1_aff_raw(P1), util.g2_aff_raw(Q1), util.g1_aff_raw(P2)) == 1
    # round 3: the two line values of a step merged before they touch f (k_lines2s / k_millerf2s), fixed lines normalised so that
    # the w^3 coefficient is yP: same final exponentiation as the plain loops, both tables, both tower instantiations
    assert hs.hs_miller_merged_matches(util.g1_aff_raw(P1), util.g2_aff_raw(Q1), util.g1_aff_raw(P2), util.g2_aff_raw(Q2)) == 1


def test_pairing_product_by_entries_and_horner_chain(hs):
    """The pairing product as run_miller_product_tree takes it (csrc/kernels.cuh k_linesp / k_line_quad / k_f12_fold4, engine program
    HORNER): per Miller entry the product over the items of their line values (four items merged two by two and multiplied, then folds
    by fours), ONE Horner chain over the 68 entry products -- on the lane-split tower with the bound tracker on, against the product
    of the plain Miller loops after the final exponentiation.  Item counts that leave quads and fold levels partly empty (absent
    items enter as the line 1); with the last pair's line values from the normalised table of -[c] g2 (k_lines_fixed)."""
    rng = random.Random(11)
    cinv = pow(c.H_EFF_G1, -1, c.R)
    negc = c.E2.neg(c.E2.mul(c.G2_GEN, cinv))
    for n, fixed in ((1, 0), (4, 0), (5, 1), (6, 0), (17, 1), (21, 0)):
        ps = [c.E1.mul(c.G1_GEN, rng.randrange(1, c.R)) for _ in range(n)]
        qs = [c.E2.mul(c.G2_GEN, rng.randrange(1, c.R)) for _ in range(n)]
        if fixed:
            qs[-1] = negc
        g1 = b''.join(util.g1_aff_raw(p) for p in ps)
        g2 = b''.join(util.g2_aff_raw(q) for q in qs)
        assert hs.hs_product_tree_matches(n, g1, g2, fixed) == 1, (n, fixed)


def test_verify_items(hs):
    """core_verify per item (reference src/traits/sig_core.rs:120-146): verdicts and error precedence, both backends,
    all schemes, plus the C++ known-answer signatures (tests/cpp_integration_test.rs:54-82)."""
    rng = random.Random(5)
    k = json.load(open(os.path.join(util.ROOT, 'tests', 'golden', 'ref_kats.json')))['cpp']
    C = ref.G2Impl
    msg = bytes.fromhex(k['message'])
    dst = C.DST[ref.BASIC]
    for pk_h, sig_h in zip(k['pk'], k['sig']):
        pk, sig = C.pk_from_bytes(bytes.fromhex(pk_h)), C.sig_from_bytes(bytes.fromhex(sig_h))
        assert hs.hs_verify(2, util.g1_raw(pk, rng), util.g2_raw(sig, rng), 0, msg, len(msg), dst, len(dst)) == 0
        assert hs.hs_verify(2, util.g1_raw(pk, rng), util.g2_raw(sig, rng), 0, b'hellp', 5, dst, len(dst)) == 1
    # Bls12381G1Impl twice: as the kernels verify (message point uncleared, second pair (sig, -[c] g2), its own line table) and
    # in the textbook form (cleared hash, -g2): the same verdicts
    no_clear = ctypes.c_int.in_dll(hs, 'hs_no_clear')
    merged = ctypes.c_int.in_dll(hs, 'hs_merged_lines')   # the lane-split leg: 1 = the two-kernel loop over merged lines (default)
    for C, sg, nc, mg in ((ref.G1Impl, 1, 1, 1), (ref.G1Impl, 1, 0, 1), (ref.G2Impl, 2, 1, 1), (ref.G1Impl, 1, 1, 0), (ref.G2Impl, 2, 1, 0)):
        no_clear.value = nc
        merged.value = mg
        pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
        for sch in (ref.BASIC, ref.AUG, ref.POP):
            sk = ref.keygen_from_hash(bytes([sch + 7 * sg]) * 32)
            pk = ref.public_key(C, sk)
            m = b'signatures_work'
            sig = ref.sign(C, sch, sk, m)
            d = C.DST[sch]
            aug = int(sch == ref.AUG)
            assert hs.hs_verify(sg, pkraw(pk, rng), sigraw(sig, rng), aug, m, len(m), d, len(d)) == 0
            assert hs.hs_verify(sg, pkraw(pk, rng), sigraw(sig, rng), aug, b'bad', 3, d, len(d)) == 1
            assert hs.hs_verify(sg, pkraw(pk, rng), sigraw(None), aug, m, len(m), d, len(d)) == 2
            assert hs.hs_verify(sg, pkraw(None), sigraw(sig, rng), aug, m, len(m), d, len(d)) == 3
            assert hs.hs_verify(sg, pkraw(None), sigraw(None), aug, m, len(m), d, len(d)) == 2
    # a RAW signature with a cofactor-torsion component (T1 = [r] R1 lies in r E1(Fp), where the reduced pairing is trivial): the same
    # verdict in the cleared form (pair (sig, -g2)) and in the uncleared one the kernels run (pair (sig, -[c] g2)); a RAW key outside G2
    # is never accepted (advisor's low finding / VERDICT r2 weak #1; the GPU twin is tests/test_gpu_round2.py)
    C = ref.G1Impl
    while True:
        x = rng.randrange(P)
        y = c.fp_sqrt(c.E1.rhs(x))
        T1 = c.E1.mul((x, y), c.R) if y is not None else None
        if T1 is not None:
            break
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        y = c.f2_sqrt(c.E2.rhs(x))
        T2 = c.E2.mul((x, y), c.R) if y is not None else None
        if T2 is not None:
            break
    sk, m, d = 424242, b'torsion', C.DST[ref.POP]
    pk, sig = ref.public_key(C, sk), ref.sign(C, ref.POP, sk, m)
    for nc in (1, 0):
        for mg in (1, 0):
            no_clear.value, merged.value = nc, mg
            assert hs.hs_verify(1, util.g2_raw(pk, rng), util.g1_raw(c.E1.add(sig, T1), rng), 0, m, len(m), d, len(d)) == 0
            assert hs.hs_verify(1, util.g2_raw(c.E2.add(pk, T2), rng), util.g1_raw(sig, rng), 0, m, len(m), d, len(d)) == 1
    no_clear.value = 1
    merged.value = 1


def test_decompress(hs):
    """Checked decompression (device code on the host) vs the oracle: sign bits, infinity, legacy, malformed, off-curve and
    on-curve-but-not-in-subgroup encodings (fast endomorphism subgroup tests vs the oracle's [r]P)."""
    rng = random.Random(12)

    def dec(group, b, legacy=0):
        out = ctypes.create_string_buffer(48 * group)
        rc = hs.hs_decompress(group, b, legacy, out)
        return rc, out.raw
    for group, E, gen, comp in ((1, c.E1, c.G1_GEN, c.g1_compress), (2, c.E2, c.G2_GEN, c.g2_compress)):
        w = 48 * group
        for _ in range(4):
            pt = E.mul(gen, rng.randrange(1, c.R))
            b = comp(pt)
            assert dec(group, b) == (0, b) and dec(group, ref.modern_to_legacy(b), 1) == (0, b)
            assert dec(group, bytes([b[0] ^ 0x20]) + b[1:]) == (0, comp(E.neg(pt)))
        assert dec(group, comp(None)) == (0, comp(None)) and dec(group, comp(None), 1) == (0, comp(None))
        for bb in (bytes(w), bytes([0x40]) + bytes(w - 1), bytes([0xc0]) + bytes(w - 2) + b'\x01', bytes([0xe0]) + bytes(w - 1),
                   bytes([0x9f]) + b'\xff' * (w - 1)):
            assert dec(group, bb)[0] == 7
        assert dec(group, bytes([0xff]) * w, 1)[0] == 8 and dec(group, bytes([0x20]) + bytes(w - 1), 1)[0] == 8
        x0 = rng.randrange(c.P)
        seen = set()
        for k in range(30):
            x = (x0 + k) % c.P if group == 1 else ((x0 + k) % c.P, 5)
            y = c.fp_sqrt(E.rhs(x)) if group == 1 else c.f2_sqrt(E.rhs(x))
            enc = x.to_bytes(48, 'big') if group == 1 else x[1].to_bytes(48, 'big') + x[0].to_bytes(48, 'big')
            rc, _ = dec(group, bytes([enc[0] | 0x80]) + enc[1:])
            if y is None:
                assert rc == 7
                seen.add('off')
            else:
                insub = c.g1_in_subgroup((x, y)) if group == 1 else c.g2_in_subgroup((x, y))
                assert (rc == 0) == insub
                seen.add('sub' if insub else 'nosub')
        assert {'off', 'nosub'} <= seen


def test_msm2_per_item_functions(hs):
    """csrc/msm2.cuh on the host with the bound tracker on: the base-z (G2) and base-z^2 (G1) scalar decompositions, the four
    (two) endomorphism images with the decomposition's signs, the mixed addition incl. its exceptional cases on both towers,
    and the whole signed-window bucket evaluation for several window widths -- against the oracle's plain sum of k_i P_i
    (the loop of reference src/secure_aggregation.rs:201-204)."""
    rng = random.Random(21)
    z = 0xd201000000010000
    a = (ctypes.c_uint64 * 4)()
    ks = [0, 1, z - 1, z, z * z, z ** 3, c.R - 1, c.R, c.R + 7, 2 ** 255, 2 ** 256 - 1] + [rng.randrange(2 ** 256) for _ in range(40)]
    for k in ks:
        kb = k.to_bytes(32, 'little')
        hs.hs_msm2_decompose(2, kb, a)
        assert all(x < z for x in a) and sum(int(x) * z ** j for j, x in enumerate(a)) == k % c.R
        hs.hs_msm2_decompose(1, kb, a)
        a0, a1 = int(a[0]) | int(a[1]) << 64, int(a[2]) | int(a[3]) << 64
        assert a0 < z * z and a0 + a1 * z * z == k % c.R
    # images
    P1, P2 = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R)), c.E2.mul(c.G2_GEN, rng.randrange(1, c.R))
    out = ctypes.create_string_buffer(4 * 96)
    hs.hs_msm2_images(1, util.g1_raw(P1, rng), out)
    assert out.raw[:48] == c.g1_compress(P1) and out.raw[48:96] == c.g1_compress(c.E1.mul(P1, z * z % c.R))   # -phi(P) = [z^2] P
    hs.hs_msm2_images(2, util.g2_raw(P2, rng), out)
    for j in range(4):
        assert out.raw[96 * j:96 * (j + 1)] == c.g2_compress(c.E2.mul(P2, pow(z, j, c.R))), j                   # Q_j = [z^j] P
    # mixed addition: generic, accumulator at infinity, equal points (doubling), opposite points (infinity)
    o1, o2 = ctypes.create_string_buffer(96), ctypes.create_string_buffer(96)
    for group, E, raw, comp, P in ((1, c.E1, util.g1_raw, c.g1_compress, P1), (2, c.E2, util.g2_raw, c.g2_compress, P2)):
        Q = E.mul(P, 12345)
        for acc, q, neg in ((P, Q, 0), (P, Q, 1), (None, Q, 0), (None, Q, 1), (Q, Q, 0), (Q, Q, 1), (E.neg(Q), Q, 0)):
            hs.hs_msm2_madd(group, raw(acc, rng), raw(q, rng), neg, o1, o2)
            want = comp(E.add(acc, E.neg(q) if neg else q))
            assert o1.raw[:48 * group] == want, (group, neg)
            if group == 2:
                assert o2.raw == want
    # whole evaluation
    for group, E, gen, raw, comp in ((1, c.E1, c.G1_GEN, util.g1_raw, c.g1_compress), (2, c.E2, c.G2_GEN, util.g2_raw, c.g2_compress)):
        n = 9
        pts = [E.mul(gen, rng.randrange(1, c.R)) for _ in range(n)]
        pts[3] = None
        pts[5] = pts[1]
        scal = [rng.randrange(c.R) for _ in range(n)]
        scal[0], scal[2], scal[4] = 0, c.R - 1, 2 ** 256 - 1
        want = None
        for p, s in zip(pts, scal):
            want = E.add(want, E.mul(p, s % c.R))
        blob = b''.join(raw(p, rng) for p in pts)
        sb = b''.join(s.to_bytes(32, 'little') for s in scal)
        for W in (9, 10, 13, 22):           # window counts that divide the bits evenly and unevenly (widths differ by one)
            hs.hs_msm2_small(group, n, blob, sb, W, o1)
            assert o1.raw[:48 * group] == comp(want), (group, W)


def test_device_headers_under_ubsan(tmp_path):
    """The device arithmetic headers as host C++ once more with -fsanitize=undefined -fno-sanitize-recover (signed overflow of the
    lazy limbs, shifts, misaligned accesses): the pairing and verification tests of this file in a child process on that build."""
    import subprocess
    import sys
    so = str(tmp_path / 'libhostsim_ubsan.so')
    subprocess.check_call(['g++', '-O1', '-g', '-DBLS_TRACK_BOUNDS', '-fsanitize=undefined', '-fno-sanitize-recover=undefined', '-shared', '-fPIC', '-o', so,
                           os.path.join(util.ROOT, 'tests', 'hostsim', 'hostsim.cpp')])
    env = dict(os.environ, BLS_HOSTSIM_SO=so)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(util.ROOT, 'tests', 'test_hostsim.py'), '-q', '-x', '-p', 'no:cacheprovider',
                        '-k', 'fp_ops or fp12_ops or pairing_values or verify_items or g2_split or msm2 or karatsuba'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
