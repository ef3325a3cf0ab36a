"""CPU restatement of the reference's verify-side algorithms -- ORACLE (test infrastructure).

Follows, function by function, the control flow and error order of dashpay/agora-blsful:

  core_verify / core_aggregate_verify / aggregate_public_keys   src/traits/sig_core.rs:50-59,120-178
  Basic / MessageAugmentation / ProofOfPossession schemes       src/traits/sig_basic.rs:36-64,
                                                                src/traits/sig_aug.rs:20-47, src/traits/sig_pop.rs:37-70
  pairing glue (argument order, to_affine)                      src/helpers.rs:41-63, src/impls/g1.rs:38-40, g2.rs:36-38
  DSTs                                                          src/impls/g1.rs:110-119, src/impls/g2.rs:108-117
  secure aggregation (sort, SHA-256 coefficients, sum t_i pk_i) src/secure_aggregation.rs:37-106,173-256,269-425
  legacy header transcode                                       src/impls/legacy.rs:19-82
  keygen used only to make test inputs                          src/helpers.rs:9-26, src/secret_key.rs:276-281
  signature proof of knowledge, signcryption validity / share   src/traits/sig_proof.rs:14-142, src/traits/sign_crypt.rs:69-77,
                                                                153-160,192-207

Arithmetic comes from oracle/py/bls381.py.  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import hashlib
import hmac

from . import bls381 as c

BASIC, AUG, POP = 0, 1, 2                 # SignatureSchemes, src/sig_types.rs:6-13
MODERN, LEGACY = 0, 1                     # SerializationFormat, src/serialization.rs:11-17


class BlsError(Exception):
    """Mirror of src/error.rs:5-55 (only the variants reachable on the verify path)."""

    def __init__(self, kind, msg=''):
        super().__init__(f'{kind}: {msg}' if msg else kind)
        self.kind, self.msg = kind, msg

    def __eq__(self, o):
        return isinstance(o, BlsError) and (self.kind, self.msg) == (o.kind, o.msg)

    def __hash__(self):
        return hash((self.kind, self.msg))


def InvalidInputs(m):
    return BlsError('InvalidInputs', m)


InvalidSignature = BlsError('InvalidSignature')
InvalidCoefficient = BlsError('InvalidCoefficient')
InvalidProof = BlsError('InvalidProof')


class _Impl:
    pass


class G1Impl(_Impl):
    """Bls12381G1Impl: signatures in G1 (48 B), public keys in G2 (96 B).  src/impls/g1.rs"""
    name = 'G1'
    DST = {BASIC: b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_',
           AUG: b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_AUG_',
           POP: b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_'}
    POP_DST = b'BLS_POP_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_'
    sig_curve, pk_curve = c.E1, c.E2
    pk_gen = c.G2_GEN
    hash_to_point = staticmethod(c.hash_to_g1)
    pk_to_bytes = staticmethod(c.g2_compress)
    pk_from_bytes = staticmethod(c.g2_decompress)
    sig_to_bytes = staticmethod(c.g1_compress)
    sig_from_bytes = staticmethod(c.g1_decompress)
    PK_BYTES, SIG_BYTES = 96, 48

    @staticmethod
    def pairing_is_identity(pairs):
        """pairs: [(Signature-group point, PublicKey-group point)]; src/helpers.rs:41-51."""
        return c.pairing_product_is_one([(s, p) for (s, p) in pairs])


class G2Impl(_Impl):
    """Bls12381G2Impl: signatures in G2 (96 B), public keys in G1 (48 B).  src/impls/g2.rs"""
    name = 'G2'
    DST = {BASIC: b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_',
           AUG: b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_AUG_',
           POP: b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_'}
    POP_DST = b'BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_'
    sig_curve, pk_curve = c.E2, c.E1
    pk_gen = c.G1_GEN
    hash_to_point = staticmethod(c.hash_to_g2)
    pk_to_bytes = staticmethod(c.g1_compress)
    pk_from_bytes = staticmethod(c.g1_decompress)
    sig_to_bytes = staticmethod(c.g2_compress)
    sig_from_bytes = staticmethod(c.g2_decompress)
    PK_BYTES, SIG_BYTES = 48, 96

    @staticmethod
    def pairing_is_identity(pairs):
        """src/helpers.rs:53-63: arguments swapped so G1 is always the first pairing argument."""
        return c.pairing_product_is_one([(p, s) for (s, p) in pairs])


# ------------------------------------------------------------------ src/traits/sig_core.rs
def core_verify(C, pk, sig, msg, dst):
    """sig_core.rs:120-146.  Order: sig identity, pk identity, pairing."""
    if sig is None:
        raise InvalidInputs('signature is the identity point')
    if pk is None:
        raise InvalidInputs('public key is the identity point')
    a = C.hash_to_point(msg, dst)
    neg_g = C.pk_curve.neg(C.pk_gen)
    if C.pairing_is_identity([(a, pk), (sig, neg_g)]):
        return
    raise InvalidSignature


def core_aggregate_verify(C, items, sig, dst):
    """sig_core.rs:149-178.  items: iterable of (pk, msg)."""
    if sig is None:
        raise InvalidInputs('signature is the identity point')
    pairs = []
    for i, (pk, msg) in enumerate(items):
        if pk is None:
            raise InvalidInputs('public key at %d is the identity point' % (i + 1))
        pairs.append((C.hash_to_point(msg, dst), pk))
    pairs.append((sig, C.pk_curve.neg(C.pk_gen)))
    if C.pairing_is_identity(pairs):
        return
    raise InvalidSignature


def aggregate_public_keys(C, pks):
    """sig_core.rs:50-59 == src/traits/pk_multi.rs:7-13: serial point sum."""
    r = None
    for p in pks:
        r = C.pk_curve.add(r, p)
    return r


def aggregate_signatures(C, sigs):
    r = None
    for s in sigs:
        r = C.sig_curve.add(r, s)
    return r


# ------------------------------------------------------------------ schemes
def verify(C, scheme, pk, sig, msg):
    """Signature::verify, src/signature.rs:130-138 -> sig_basic.rs:36, sig_aug.rs:20-24, sig_pop.rs:37."""
    if scheme == AUG:
        msg = C.pk_to_bytes(pk) + msg
    return core_verify(C, pk, sig, msg, C.DST[scheme])


def aggregate_verify(C, scheme, items, sig):
    """AggregateSignature::verify, src/aggregate_signature.rs:230-239."""
    items = list(items)
    if scheme == BASIC:                                   # sig_basic.rs:41-64 duplicate check first
        seen = {}
        for i, (_, m) in enumerate(items):
            m = bytes(m)
            if m in seen:
                raise InvalidInputs('duplicate messages detected at %d and %d' % (seen[m], i))
            seen[m] = i
    elif scheme == AUG:                                   # sig_aug.rs:27-38
        items = [(pk, (C.pk_to_bytes(pk) if pk is not None else C.pk_to_bytes(None)) + m) for pk, m in items]
    return core_aggregate_verify(C, items, sig, C.DST[scheme])


def multi_sig_verify(C, scheme, pks, sig, msg):
    """MultiPublicKey::from_public_keys (src/multi_public_key.rs:79-83) + MultiSignature::verify
    (src/multi_signature.rs:127-135)."""
    return verify(C, scheme, aggregate_public_keys(C, pks), sig, msg)


def pop_verify(C, pk, sig):
    """sig_pop.rs:67-70."""
    return core_verify(C, pk, sig, C.pk_to_bytes(pk), C.POP_DST)


# ------------------------------------------------------------------ src/traits/sig_proof.rs
def sig_proof_generate(C, sig, msg, dst, x, y):
    """generate_commitment (sig_proof.rs:15-26) + generate_proof (:49-72) with the random x supplied by the caller:
    (U, V) = (x * H(msg), -(x + y) * sig)."""
    if x % c.R == 0:
        raise InvalidInputs('x is the zero')
    u = C.sig_curve.mul(C.hash_to_point(msg, dst), x)
    if u is None:
        raise InvalidInputs('commitment is the identity point')
    if sig is None:
        raise InvalidInputs('signature is the identity point')
    if y % c.R == 0:
        raise InvalidInputs('y is the zero')
    return u, C.sig_curve.neg(C.sig_curve.mul(sig, (x + y) % c.R))


def sig_proof_verify(C, commitment, proof, pk, y, msg, dst):
    """BlsSignatureProof::verify, sig_proof.rs:102-142.  Order: commitment, proof, pk identity; y zero; pairing."""
    if commitment is None:
        raise InvalidInputs('commitment is the identity point')
    if proof is None:
        raise InvalidInputs('proof is the identity point')
    if pk is None:
        raise InvalidInputs('pk is the identity point')
    if y % c.R == 0:
        raise InvalidInputs('y is the zero')
    a = C.hash_to_point(msg, dst)
    t = C.sig_curve.add(commitment, C.sig_curve.mul(a, y))
    if C.pairing_is_identity([(proof, C.pk_gen), (t, pk)]):
        return
    raise InvalidProof


# ------------------------------------------------------------------ src/traits/sign_crypt.rs
def signcrypt_compute_w(C, u, v, dst):
    """compute_w, sign_crypt.rs:153-160: H(U.to_bytes() || V)."""
    return C.hash_to_point(C.pk_to_bytes(u) + bytes(v), dst)


def signcrypt_valid(C, u, v, w, dst):
    """BlsSignCrypt::valid, sign_crypt.rs:69-77 (returns the Choice as a bool)."""
    if u is None or w is None:                 # the Choice is masked by !u.is_identity() & !w.is_identity()
        return False
    w_tick = signcrypt_compute_w(C, u, v, dst)
    g = C.pk_curve.neg(C.pk_gen)
    return C.pairing_is_identity([(w, g), (w_tick, u)])


def signcrypt_verify_share(C, share, pk, u, v, w, dst):
    """BlsSignCrypt::verify_share, sign_crypt.rs:192-207."""
    if share is None or pk is None or w is None:
        return False
    h = C.sig_curve.neg(signcrypt_compute_w(C, u, v, dst))
    return C.pairing_is_identity([(h, share), (w, pk)])


# ------------------------------------------------------------------ src/impls/legacy.rs
def modern_to_legacy(b):
    """legacy.rs:19-35."""
    b = bytearray(b)
    if b[0] == 0xc0:
        return bytes(b)
    ysign = b[0] & 0x20
    b[0] &= 0x1f
    if ysign:
        b[0] |= 0x80
    return bytes(b)


def legacy_to_modern(b):
    """legacy.rs:39-67."""
    b = bytearray(b)
    if b[0] == 0xc0:
        return bytes(b)
    orig = b[0]
    ysign = b[0] & 0x80
    b[0] &= 0x7f
    if b[0] & 0xe0:
        raise BlsError('LegacyFormatError',
                       'Invalid legacy format: unexpected bits in byte[0] = 0x%02x' % orig)
    b[0] |= 0x80
    if ysign:
        b[0] |= 0x20
    return bytes(b)


def validate_modern(b0, kind):
    """legacy.rs:71-82."""
    if b0 != 0xc0 and (b0 & 0xc0) != 0x80:
        raise BlsError('DeserializationError',
                       'Invalid modern %s format: byte[0] = 0x%02x, expected bit pattern 10xxxxxx' % (kind, b0))


def pk_to_bytes_with_mode(C, pk, fmt):
    """PublicKey::to_bytes_with_mode, src/public_key.rs:146-151 (48-byte keys only => G2Impl)."""
    b = C.pk_to_bytes(pk)
    return modern_to_legacy(b) if fmt == LEGACY else b


def pk_from_bytes_with_mode(C, b, fmt):
    """src/public_key.rs:158-171 -> legacy.rs:100-126."""
    if len(b) != 48:
        raise BlsError('InvalidLength', 'expected 48, actual %d' % len(b))
    try:
        if fmt == MODERN:
            validate_modern(b[0], 'G1')
            return c.g1_decompress(b)
        return c.g1_decompress(legacy_to_modern(b))
    except c.DecodeError:
        raise BlsError('DeserializationError',
                       'Invalid G1 point' + (' after conversion' if fmt == LEGACY else ''))


def sig_from_bytes_with_mode(C, b, fmt):
    """Signature::from_bytes_with_mode, src/signature.rs:231-253 (96-byte G2 signatures)."""
    if len(b) != 96:
        raise BlsError('InvalidLength', 'expected 96, actual %d' % len(b))
    try:
        if fmt == MODERN:
            validate_modern(b[0], 'G2')
            return c.g2_decompress(b)
        return c.g2_decompress(legacy_to_modern(b))
    except c.DecodeError:
        raise BlsError('DeserializationError',
                       'Invalid G2 point' + (' after conversion' if fmt == LEGACY else ''))


# ------------------------------------------------------------------ src/secure_aggregation.rs
def secure_coefficients(key_bytes):
    """The SHA-256 part of hash_public_keys_with_sorted (:37-106) on already-serialised keys.

    Returns (perm, H, [t_i]) with perm[i] = index into the input of the i-th sorted key.
    Rust's sort_by is stable; Python's sorted() is too.  t_i = int_BE(SHA256(BE32(i) || H)) mod r
    (the reference's own known-answer tests only pass with reduce semantics)."""
    perm = sorted(range(len(key_bytes)), key=lambda i: key_bytes[i])
    h = hashlib.sha256()
    for i in perm:
        h.update(key_bytes[i])
    H = h.digest()
    ts = []
    for i in range(len(perm)):
        t = int.from_bytes(hashlib.sha256(i.to_bytes(4, 'big') + H).digest(), 'big') % c.R
        if t == 0:
            raise InvalidCoefficient
        ts.append(t)
    return perm, H, ts


def verify_secure(C, scheme, pks, sig, msg, fmt=None):
    """Signature::verify_secure (src/signature.rs:177-197) / verify_secure_with_mode (:256-276)
    -> verify_secure_with_dst_internal (src/secure_aggregation.rs:173-208).
    Note (as in the reference :236-246) the Aug scheme only switches the DST here."""
    if len(pks) == 0:
        if sig is None:
            return
        raise InvalidSignature
    if fmt is None:
        kb = [C.pk_to_bytes(p) for p in pks]
    else:
        kb = [pk_to_bytes_with_mode(C, p, fmt) for p in pks]
    perm, _, ts = secure_coefficients(kb)
    agg = None
    for i, t in zip(perm, ts):
        agg = C.pk_curve.add(agg, C.pk_curve.mul(pks[i], t))
    return core_verify(C, agg, sig, msg, C.DST[scheme])


def aggregate_secure(C, pks, sigs, fmt=None):
    """Sign-side aggregate_secure_internal (:110-169); used to build test inputs.  Reproduces the
    reference's first-match index search (duplicate keys pick the first matching signature)."""
    if len(pks) != len(sigs):
        raise InvalidInputs('Mismatched array lengths')
    if not pks:
        return None
    ser = (lambda p: C.pk_to_bytes(p)) if fmt is None else (lambda p: pk_to_bytes_with_mode(C, p, fmt))
    kb = [ser(p) for p in pks]
    perm, _, ts = secure_coefficients(kb)
    agg = None
    for i, t in zip(perm, ts):
        idx = kb.index(kb[i])
        agg = C.sig_curve.add(agg, C.sig_curve.mul(sigs[idx], t))
    return agg


# ------------------------------------------------------------------ test-input generation (sign side)
def keygen_from_hash(seed):
    """SecretKey::from_hash, src/secret_key.rs:276-281 -> scalar_from_hkdf_bytes, src/helpers.rs:9-26."""
    salt = b'BLS-SIG-KEYGEN-SALT-'
    prk = hmac.new(salt, seed + b'\x00', hashlib.sha256).digest()
    okm, t, i = b'', b'', 1
    while len(okm) < 48:
        t = hmac.new(prk, t + bytes([0, 48]) + bytes([i]), hashlib.sha256).digest()
        okm += t
        i += 1
    s = int.from_bytes(okm[:48], 'big') % c.R
    assert s != 0
    return s


def public_key(C, sk):
    return C.pk_curve.mul(C.pk_gen, sk)


def sign(C, scheme, sk, msg):
    """core_sign, sig_core.rs:108-117 (+ sig_aug.rs augmentation)."""
    if scheme == AUG:
        msg = C.pk_to_bytes(public_key(C, sk)) + msg
    return C.sig_curve.mul(C.hash_to_point(msg, C.DST[scheme]), sk)


def pop_prove(C, sk):
    return C.sig_curve.mul(C.hash_to_point(C.pk_to_bytes(public_key(C, sk)), C.POP_DST), sk)
