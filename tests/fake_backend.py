"""TEST-ONLY stand-in for agora-blsful_amd/api.py backed by the oracle, so that the sharding/orchestration logic of
agora-blsful_amd/dist.py can be exercised with world_size 2 over gloo on a machine without a GPU.  The product never
imports this."""
import util
from util import c, ref

OK, INVALID_SIGNATURE, SIG_IDENTITY, PK_IDENTITY = 0, 1, 2, 3
DST = {(1, s): ref.G1Impl.DST[s] for s in (0, 1, 2)}
DST.update({(2, s): ref.G2Impl.DST[s] for s in (0, 1, 2)})
_TOWER_TO_W = [0, 2, 4, 1, 3, 5]          # record slot (c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2) -> power of w


def g1_from_raw(b):
    x, y, z = (util.fp_from_raw(b[48 * i:48 * i + 48]) for i in range(3))
    if z == 0:
        return None
    zi = pow(z, -1, c.P)
    return (x * zi * zi % c.P, y * zi * zi * zi % c.P)


def g2_from_raw(b):
    v = [(util.fp_from_raw(b[96 * i:96 * i + 48]), util.fp_from_raw(b[96 * i + 48:96 * i + 96])) for i in range(3)]
    if v[2] == (0, 0):
        return None
    zi = c.f2_inv(v[2])
    z2 = c.f2_sqr(zi)
    return (c.f2_mul(v[0], z2), c.f2_mul(v[1], c.f2_mul(z2, zi)))


def _impl(sg):
    return ref.G1Impl if sg == 1 else ref.G2Impl


def _pk(sg, b):
    return g2_from_raw(b) if sg == 1 else g1_from_raw(b)


def _sig(sg, b):
    return g1_from_raw(b) if sg == 1 else g2_from_raw(b)


def _status(fn):
    try:
        fn()
        return OK
    except ref.BlsError as e:
        if e.kind == 'InvalidSignature':
            return INVALID_SIGNATURE
        return SIG_IDENTITY if 'signature' in e.msg else PK_IDENTITY


def verify_batch(sg, scheme, pks, sigs, msgs):
    C = _impl(sg)
    return [_status(lambda: ref.verify(C, scheme, _pk(sg, p), _sig(sg, s), m)) for p, s, m in zip(pks, sigs, msgs)]


def core_verify(sg, dst, pks, sigs, msgs):
    C = _impl(sg)
    return [_status(lambda: ref.core_verify(C, _pk(sg, p), _sig(sg, s), m, dst)) for p, s, m in zip(pks, sigs, msgs)]


def pop_verify_batch(sg, pks, proofs):
    C = _impl(sg)
    return [_status(lambda: ref.pop_verify(C, _pk(sg, p), _sig(sg, s))) for p, s in zip(pks, proofs)]


def sig_proof_verify_batch(sg, scheme, us, vs, pks, ys, msgs):
    C = _impl(sg)
    codes = {'commitment is the identity point': 9, 'proof is the identity point': 10, 'pk is the identity point': 3, 'y is the zero': 11}
    out = []
    for u, v, p, y, m in zip(us, vs, pks, ys, msgs):
        try:
            ref.sig_proof_verify(C, _sig(sg, u), _sig(sg, v), _pk(sg, p), y, m, C.DST[scheme])
            out.append(OK)
        except ref.BlsError as e:
            out.append(INVALID_SIGNATURE if e.kind == 'InvalidProof' else codes[e.msg])
    return out


def signcrypt_valid_batch(sg, scheme, us, ws, vs):
    C = _impl(sg)
    return [ref.signcrypt_valid(C, _pk(sg, u), v, _sig(sg, w), C.DST[scheme]) for u, w, v in zip(us, ws, vs)]


def point_sum(group, pts, scalars=None):
    E, dec, enc = (c.E1, g1_from_raw, util.g1_raw) if group == 1 else (c.E2, g2_from_raw, util.g2_raw)
    acc = None
    for i, b in enumerate(pts):
        p = dec(b)
        acc = E.add(acc, p if scalars is None else E.mul(p, scalars[i]))
    return enc(acc)


def serialize(group, pts, legacy=False):
    out = [c.g1_compress(g1_from_raw(b)) if group == 1 else c.g2_compress(g2_from_raw(b)) for b in pts]
    return [ref.modern_to_legacy(b) for b in out] if legacy else out


def secure_coefficients(kb):
    try:
        perm, _, ts = ref.secure_coefficients(kb)
        return 0, perm, ts
    except ref.BlsError:
        return 5, [], []


def aggregate_partial(sg, scheme, pks, msgs, sig=None):
    C = _impl(sg)
    P = [_pk(sg, b) for b in pks]
    s = _sig(sg, sig) if sig is not None else 1
    fb = -1
    if sig is not None and s is None:
        fb = len(pks)
    else:
        for i, p in enumerate(P):
            if p is None:
                fb = i
                break
    if fb >= 0:
        return util.f12_record(c.F12_ONE), fb
    pairs = []
    for p, m in zip(P, msgs):
        if scheme == 1:
            m = C.pk_to_bytes(p) + m
        h = C.hash_to_point(m, C.DST[scheme])
        pairs.append((h, p) if sg == 1 else (p, h))
    if sig is not None:
        neg = C.pk_curve.neg(C.pk_gen)
        pairs.append((s, neg) if sg == 1 else (neg, s))
    return util.f12_record(c.miller_loop(pairs)), fb


def fp12_product_is_one(records):
    f = c.F12_ONE
    for r in records:
        f = c.f12_mul(f, util.f12_from_record(r))
    return c.final_exponentiation(f) == c.F12_ONE
