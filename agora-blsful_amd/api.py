"""Host-side mirror of the reference's verify-path interface over the C ABI of libblsgpu.so (include/blsgpu.h).

The reference is a Rust crate and no Rust toolchain exists in this image, so this module plays the role of the
`hip`-feature shim of INTEGRATION.md: same names, argument meaning and error behaviour as

    Signature::verify / verify_secure / verify_secure_with_mode     reference src/signature.rs:130-138,177-197,256-276
    MultiSignature::verify + MultiPublicKey::from_public_keys      src/multi_signature.rs:127-135, src/multi_public_key.rs:79-83
    AggregateSignature::verify                                     src/aggregate_signature.rs:230-239
    BlsError                                                       src/error.rs:5-55

so that the parity tests read like the reference's own tests.  Points are held as RAW_PROJ byte strings (the
blst in-memory layout the Rust types wrap).  All compute happens in the HIP library; if it cannot be loaded or finds
no gfx950 device, every call raises -- there is no CPU path here.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libblsgpu.so')

FMT_RAW_PROJ, FMT_RAW_AFFINE, FMT_COMPRESSED, FMT_LEGACY = 0, 1, 2, 3
BASIC, AUG, POP = 0, 1, 2
MODERN, LEGACY = 0, 1

OK, INVALID_SIGNATURE, SIG_IDENTITY, PK_IDENTITY, DUPLICATE_MESSAGE, INVALID_COEFFICIENT, BAD_LENGTH, BAD_ENCODING, \
    LEGACY_FORMAT = range(9)
COMMITMENT_IDENTITY, PROOF_IDENTITY, ZERO_CHALLENGE = 9, 10, 11    # blsgpu_sig_proof_verify_batch only

EXPORTS = [
    'blsgpu_init', 'blsgpu_shutdown', 'blsgpu_last_error', 'blsgpu_verify_batch', 'blsgpu_multi_verify',
    'blsgpu_aggregate_verify', 'blsgpu_verify_secure', 'blsgpu_secure_coefficients', 'blsgpu_hash_to_g1',
    'blsgpu_hash_to_g2', 'blsgpu_sum_g1', 'blsgpu_sum_g2', 'blsgpu_msm_g1', 'blsgpu_msm_g2',
    'blsgpu_pairing_product_is_one', 'blsgpu_serialize', 'blsgpu_sign_batch',
    'blsgpu_profile_enable', 'blsgpu_profile_count', 'blsgpu_profile_get',
    'blsgpu_aggregate_partial', 'blsgpu_fp12_product_is_one', 'blsgpu_core_verify', 'blsgpu_deserialize', 'blsgpu_pop_verify_batch', 'blsgpu_aggregate_secure',
    'blsgpu_signcrypt_valid_batch', 'blsgpu_sig_proof_verify_batch', 'blsgpu_pairing2_check_batch',
    'blsgpu_init_devices', 'blsgpu_device_count', 'blsgpu_sort_keys', 'blsgpu_sorted_keys_digest',
    'blsgpu_coefficients_for_range', 'blsgpu_first_duplicate_message', 'blsgpu_first_occurrence', 'blsgpu_core_verify_hashed', 'blsgpu_debug_wide_mul', 'blsgpu_debug_wide_program', 'blsgpu_verify_batch_grouped', 'blsgpu_signatures_from_tagged', 'blsgpu_signatures_to_tagged',
]


class BlsGpuRuntimeError(RuntimeError):
    """A negative return code of the C ABI (HIP failure, missing device, bad argument)."""


class BlsError(Exception):
    """Mirror of the reference's BlsError variants reachable on the verify path (src/error.rs:5-55)."""

    def __init__(self, kind, msg=''):
        super().__init__(f'{kind}: {msg}' if msg else kind)
        self.kind, self.msg = kind, msg

    def __eq__(self, o):
        return isinstance(o, BlsError) and (self.kind, self.msg) == (o.kind, o.msg)

    def __hash__(self):
        return hash((self.kind, self.msg))


def error_from_status(st, aux=(0, 0), aggregate=False):
    """Status code -> the exact BlsError value the reference returns (strings from src/traits/sig_core.rs:126-176,
    src/traits/sig_basic.rs:51-55, src/secure_aggregation.rs:99,193)."""
    if st == OK:
        return None
    if st == INVALID_SIGNATURE:
        return BlsError('InvalidSignature')
    if st == SIG_IDENTITY:
        return BlsError('InvalidInputs', 'signature is the identity point')
    if st == PK_IDENTITY:
        if aggregate:
            return BlsError('InvalidInputs', 'public key at %d is the identity point' % aux[0])
        return BlsError('InvalidInputs', 'public key is the identity point')
    if st == DUPLICATE_MESSAGE:
        return BlsError('InvalidInputs', 'duplicate messages detected at %d and %d' % (aux[0], aux[1]))
    if st == INVALID_COEFFICIENT:
        return BlsError('InvalidCoefficient')
    if st == BAD_LENGTH:
        return BlsError('InvalidLength')
    if st == BAD_ENCODING:
        return BlsError('DeserializationError')
    if st == LEGACY_FORMAT:
        return BlsError('LegacyFormatError')
    return BlsError('Unknown', str(st))


_lib = None


def load_library(path=None):
    """dlopen libblsgpu.so (no device needed for this).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        p = path or os.environ.get('BLSGPU_LIB') or LIB_PATH
        # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both own the device.
        # Importing torch first makes libblsgpu's DT_NEEDED libamdhip64.so.7 resolve to the copy torch already loaded.
        # (C/C++/Rust callers without torch simply get /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(p):
            raise BlsGpuRuntimeError(f'{p} not found: build it with __graft_entry__.build(); there is no CPU fallback')
        lib = ctypes.CDLL(p)
        vp, u8p, u64p, i32p, u32p, sz, ci = (ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)
        lib.blsgpu_init.argtypes = [ci]
        lib.blsgpu_shutdown.restype = None
        lib.blsgpu_last_error.argtypes = [ctypes.c_char_p, sz]
        lib.blsgpu_last_error.restype = sz
        lib.blsgpu_verify_batch.argtypes = [ci, ci, vp, vp, u8p, u64p, sz, ci, i32p]
        lib.blsgpu_multi_verify.argtypes = [ci, ci, vp, sz, vp, u8p, sz, ci, i32p]
        lib.blsgpu_aggregate_verify.argtypes = [ci, ci, vp, u8p, u64p, sz, vp, ci, i32p, u64p]
        lib.blsgpu_verify_secure.argtypes = [ci, ci, vp, sz, vp, u8p, sz, ci, ci, i32p]
        lib.blsgpu_secure_coefficients.argtypes = [u8p, sz, sz, u32p, u8p, i32p]
        lib.blsgpu_hash_to_g1.argtypes = [u8p, u64p, sz, u8p, sz, vp]
        lib.blsgpu_hash_to_g2.argtypes = [u8p, u64p, sz, u8p, sz, vp]
        for nm in ('blsgpu_sum_g1', 'blsgpu_sum_g2'):
            getattr(lib, nm).argtypes = [vp, sz, ci, vp]
        for nm in ('blsgpu_msm_g1', 'blsgpu_msm_g2'):
            getattr(lib, nm).argtypes = [vp, u8p, sz, ci, vp]
        lib.blsgpu_pairing_product_is_one.argtypes = [vp, vp, sz, ci, i32p]
        lib.blsgpu_serialize.argtypes = [ci, vp, sz, ci, ci, vp, i32p]
        lib.blsgpu_sign_batch.argtypes = [ci, ci, u8p, u8p, u64p, sz, vp, vp]
        lib.blsgpu_profile_enable.argtypes = [ci]
        lib.blsgpu_aggregate_partial.argtypes = [ci, ci, vp, u8p, u64p, sz, vp, ci, vp, ctypes.POINTER(ctypes.c_int64)]
        lib.blsgpu_fp12_product_is_one.argtypes = [vp, sz, i32p]
        lib.blsgpu_core_verify.argtypes = [ci, u8p, sz, vp, vp, u8p, u64p, sz, ci, i32p]
        lib.blsgpu_deserialize.argtypes = [ci, u8p, sz, ci, vp, i32p]
        lib.blsgpu_pop_verify_batch.argtypes = [ci, vp, vp, sz, ci, i32p]
        lib.blsgpu_aggregate_secure.argtypes = [ci, vp, vp, sz, ci, ci, vp, i32p]
        lib.blsgpu_signcrypt_valid_batch.argtypes = [ci, ci, vp, vp, u8p, u64p, sz, ci, i32p]
        lib.blsgpu_sig_proof_verify_batch.argtypes = [ci, ci, vp, vp, vp, u8p, u8p, u64p, sz, ci, i32p]
        lib.blsgpu_pairing2_check_batch.argtypes = [vp, vp, vp, vp, sz, ci, i32p]
        lib.blsgpu_profile_get.argtypes = [ci, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        lib.blsgpu_init_devices.argtypes = [ci]
        lib.blsgpu_sort_keys.argtypes = [u8p, sz, sz, u32p]
        lib.blsgpu_sorted_keys_digest.argtypes = [u8p, u32p, sz, sz, u8p]
        lib.blsgpu_coefficients_for_range.argtypes = [u8p, u32p, sz, sz, sz, u8p, i32p]
        lib.blsgpu_first_duplicate_message.argtypes = [u8p, u64p, sz, u64p]
        lib.blsgpu_first_occurrence.argtypes = [u8p, u32p, sz, sz, u32p]
        lib.blsgpu_core_verify_hashed.argtypes = [ci, vp, vp, vp, sz, i32p]
        lib.blsgpu_debug_wide_mul.argtypes = [u8p, u8p, sz, ci, u8p]
        lib.blsgpu_debug_wide_program.argtypes = [vp, sz, ci, u8p, u8p]
        lib.blsgpu_verify_batch_grouped.argtypes = [ci, ci, vp, vp, u8p, vp, sz, ci, ctypes.c_uint64, i32p]
        lib.blsgpu_signatures_from_tagged.argtypes = [ci, u8p, sz, u8p, vp, i32p]
        lib.blsgpu_signatures_to_tagged.argtypes = [ci, u8p, vp, sz, ci, u8p]
        _lib = lib
    return _lib


def _check(rc):
    if rc < 0:
        buf = ctypes.create_string_buffer(1024)
        _lib.blsgpu_last_error(buf, 1024)
        raise BlsGpuRuntimeError('libblsgpu rc=%d: %s' % (rc, buf.value.decode(errors='replace')))


def init(device=-1):
    lib = load_library()
    _check(lib.blsgpu_init(device))
    return lib


def _offsets(msgs):
    offs = (ctypes.c_uint64 * (len(msgs) + 1))()
    t = 0
    for i, m in enumerate(msgs):
        offs[i] = t
        t += len(m)
    offs[len(msgs)] = t
    return offs, b''.join(msgs)


def _ptr(b):
    """bytes / int (device address) -> c_void_p value"""
    if isinstance(b, int):
        return ctypes.c_void_p(b)
    return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)


# ------------------------------------------------------------------ flat (array) calls
def verify_batch(sig_group, scheme, pks, sigs, msgs, fmt=FMT_RAW_PROJ):
    """status list of Signature::verify for n independent (pk, msg, sig) items."""
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    st = (ctypes.c_int32 * n)()
    pkb, sgb = b''.join(pks), b''.join(sigs)
    _check(lib.blsgpu_verify_batch(sig_group, scheme, _ptr(pkb), _ptr(sgb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n, fmt,
                                   ctypes.cast(st, ctypes.c_void_p)))
    return list(st)


def verify_batch_grouped(sig_group, scheme, pks, sigs, msgs, seed=None, fmt=FMT_RAW_PROJ):
    """the OPT-IN grouped form of verify_batch (groups of eight items share one final exponentiation through a random linear
    combination; failing groups are re-verified item by item): same status list unless a group that holds an invalid item passes
    its combined check.  The 128-bit scalars are derived inside the library from the group's own inputs (include/blsgpu.h);
    `seed` is extra entropy mixed into that hash: a fresh random value by default, no secrecy needed."""
    if seed is None:
        import secrets
        seed = secrets.randbits(64)
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    st = (ctypes.c_int32 * max(n, 1))()
    pkb, sgb = b''.join(pks), b''.join(sigs)
    _check(lib.blsgpu_verify_batch_grouped(sig_group, scheme, _ptr(pkb), _ptr(sgb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n, fmt, seed,
                                           ctypes.cast(st, ctypes.c_void_p)))
    return list(st)[:n]


DST = {  # reference src/impls/g1.rs:110-119, src/impls/g2.rs:108-117
    (1, BASIC): b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_', (1, AUG): b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_AUG_',
    (1, POP): b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_', (2, BASIC): b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_',
    (2, AUG): b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_AUG_', (2, POP): b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_',
}
POP_DST = {1: b'BLS_POP_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_', 2: b'BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_'}


def core_verify(sig_group, dst, pks, sigs, msgs, fmt=FMT_RAW_PROJ):
    """status list of BlsSignatureCore::core_verify with an explicit DST (no augmentation)."""
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    st = (ctypes.c_int32 * max(n, 1))()
    pkb, sgb = b''.join(pks), b''.join(sigs)
    _check(lib.blsgpu_core_verify(sig_group, _ptr(dst), len(dst), _ptr(pkb), _ptr(sgb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n, fmt,
                                  ctypes.cast(st, ctypes.c_void_p)))
    return list(st)[:n]


def multi_verify(sig_group, scheme, pks, sig, msg, fmt=FMT_RAW_PROJ):
    lib = init()
    st = ctypes.c_int32(-99)
    pkb = b''.join(pks)
    _check(lib.blsgpu_multi_verify(sig_group, scheme, _ptr(pkb), len(pks), _ptr(sig), _ptr(msg), len(msg), fmt, ctypes.byref(st)))
    return st.value


def aggregate_verify(sig_group, scheme, pks, msgs, sig, fmt=FMT_RAW_PROJ):
    lib = init()
    offs, blob = _offsets(msgs)
    st = ctypes.c_int32(-99)
    aux = (ctypes.c_uint64 * 2)()
    pkb = b''.join(pks)
    _check(lib.blsgpu_aggregate_verify(sig_group, scheme, _ptr(pkb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), len(pks), _ptr(sig), fmt,
                                       ctypes.byref(st), ctypes.cast(aux, ctypes.c_void_p)))
    return st.value, (aux[0], aux[1])


def verify_secure(sig_group, scheme, pks, sig, msg, ser_format=MODERN, fmt=FMT_RAW_PROJ):
    lib = init()
    st = ctypes.c_int32(-99)
    pkb = b''.join(pks)
    _check(lib.blsgpu_verify_secure(sig_group, scheme, _ptr(pkb), len(pks), _ptr(sig), _ptr(msg), len(msg), ser_format, fmt, ctypes.byref(st)))
    return st.value


def secure_coefficients(key_bytes_list):
    lib = init()
    n = len(key_bytes_list)
    width = len(key_bytes_list[0]) if n else 48
    perm = (ctypes.c_uint32 * max(n, 1))()
    scal = ctypes.create_string_buffer(32 * max(n, 1))
    st = ctypes.c_int32(-99)
    blob = b''.join(key_bytes_list)
    _check(lib.blsgpu_secure_coefficients(_ptr(blob), n, width, ctypes.cast(perm, ctypes.c_void_p), ctypes.cast(scal, ctypes.c_void_p),
                                          ctypes.byref(st)))
    sc = scal.raw
    return st.value, list(perm)[:n], [int.from_bytes(sc[32 * i:32 * i + 32], 'little') for i in range(n)]


def hash_to_point(group, msgs, dst):
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    sz = 144 if group == 1 else 288
    out = ctypes.create_string_buffer(sz * n)
    fn = lib.blsgpu_hash_to_g1 if group == 1 else lib.blsgpu_hash_to_g2
    _check(fn(_ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n, _ptr(dst), len(dst), ctypes.cast(out, ctypes.c_void_p)))
    raw = out.raw
    return [raw[sz * i:sz * (i + 1)] for i in range(n)]


def point_sum(group, pts, scalars=None, fmt=FMT_RAW_PROJ):
    lib = init()
    sz = 144 if group == 1 else 288
    out = ctypes.create_string_buffer(sz)
    blob = b''.join(pts)
    if scalars is None:
        fn = lib.blsgpu_sum_g1 if group == 1 else lib.blsgpu_sum_g2
        _check(fn(_ptr(blob), len(pts), fmt, ctypes.cast(out, ctypes.c_void_p)))
    else:
        fn = lib.blsgpu_msm_g1 if group == 1 else lib.blsgpu_msm_g2
        sb = b''.join(int(s).to_bytes(32, 'little') for s in scalars)
        _check(fn(_ptr(blob), _ptr(sb), len(pts), fmt, ctypes.cast(out, ctypes.c_void_p)))
    return out.raw


def pairing_product_is_one(g1s, g2s, fmt=FMT_RAW_PROJ):
    lib = init()
    r = ctypes.c_int32(-99)
    a, b = b''.join(g1s), b''.join(g2s)
    _check(lib.blsgpu_pairing_product_is_one(_ptr(a), _ptr(b), len(g1s), fmt, ctypes.byref(r)))
    return bool(r.value)


def pop_verify_batch(sig_group, pks, proofs, fmt=FMT_RAW_PROJ):
    """status list of ProofOfPossession::verify for n (pk, proof) pairs."""
    lib = init()
    n = len(pks)
    st = (ctypes.c_int32 * max(n, 1))()
    a, b = b''.join(pks), b''.join(proofs)
    _check(lib.blsgpu_pop_verify_batch(sig_group, _ptr(a), _ptr(b), n, fmt, ctypes.cast(st, ctypes.c_void_p)))
    return list(st)[:n]


def signcrypt_valid_batch(sig_group, scheme, us, ws, vs, fmt=FMT_RAW_PROJ):
    """SignCryptCiphertext::is_valid per ciphertext (u, v, w): list of bool (the reference returns a Choice)."""
    lib = init()
    n = len(vs)
    offs, blob = _offsets(vs)
    st = (ctypes.c_int32 * max(n, 1))()
    ub, wb = b''.join(us), b''.join(ws)
    _check(lib.blsgpu_signcrypt_valid_batch(sig_group, scheme, _ptr(ub), _ptr(wb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n, fmt,
                                            ctypes.cast(st, ctypes.c_void_p)))
    return [s == OK for s in list(st)[:n]]


def proof_error_from_status(st):
    """Status of blsgpu_sig_proof_verify_batch -> the BlsError of BlsSignatureProof::verify (src/traits/sig_proof.rs:110-140)."""
    return {OK: None, INVALID_SIGNATURE: BlsError('InvalidProof'),
            COMMITMENT_IDENTITY: BlsError('InvalidInputs', 'commitment is the identity point'),
            PROOF_IDENTITY: BlsError('InvalidInputs', 'proof is the identity point'),
            PK_IDENTITY: BlsError('InvalidInputs', 'pk is the identity point'),
            ZERO_CHALLENGE: BlsError('InvalidInputs', 'y is the zero')}.get(st, BlsError('Unknown', str(st)))


def sig_proof_verify_batch(sig_group, scheme, commitments, proofs, pks, ys, msgs, fmt=FMT_RAW_PROJ):
    """status list of ProofOfKnowledge::verify(pk, msg, y) for n proofs (u, v); ys: ints or 32-byte LE scalars."""
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    st = (ctypes.c_int32 * max(n, 1))()
    ub, vb, pkb = b''.join(commitments), b''.join(proofs), b''.join(pks)
    yb = b''.join(y.to_bytes(32, 'little') if isinstance(y, int) else y for y in ys)
    _check(lib.blsgpu_sig_proof_verify_batch(sig_group, scheme, _ptr(ub), _ptr(vb), _ptr(pkb), _ptr(yb), _ptr(blob),
                                             ctypes.cast(offs, ctypes.c_void_p), n, fmt, ctypes.cast(st, ctypes.c_void_p)))
    return list(st)[:n]


def pairing2_check_batch(g1a, g2a, g1b, g2b, fmt=FMT_RAW_PROJ):
    """[Pairing::pairing(&[(a1, a2), (b1, b2)]).is_identity()] for n independent items."""
    lib = init()
    n = len(g1a)
    out = (ctypes.c_int32 * max(n, 1))()
    _check(lib.blsgpu_pairing2_check_batch(_ptr(b''.join(g1a)), _ptr(b''.join(g2a)), _ptr(b''.join(g1b)), _ptr(b''.join(g2b)), n, fmt,
                                           ctypes.cast(out, ctypes.c_void_p)))
    return [bool(x) for x in list(out)[:n]]


def aggregate_secure(sig_group, pks, sigs, ser_format=MODERN, fmt=FMT_RAW_PROJ):
    """(status, RAW_PROJ aggregate signature) of aggregate_secure[_with_mode]."""
    lib = init()
    osz = 144 if sig_group == 1 else 288
    out = ctypes.create_string_buffer(osz)
    st = ctypes.c_int32(-99)
    a, b = b''.join(pks), b''.join(sigs)
    _check(lib.blsgpu_aggregate_secure(sig_group, _ptr(a), _ptr(b), len(pks), ser_format, fmt, ctypes.cast(out, ctypes.c_void_p), ctypes.byref(st)))
    return st.value, out.raw


def deserialize(group, blobs, legacy=False):
    """(RAW_PROJ points, statuses) from 48/96-byte encodings: checked decompression on the GPU."""
    lib = init()
    n = len(blobs)
    osz = 144 if group == 1 else 288
    out = ctypes.create_string_buffer(osz * max(n, 1))
    st = (ctypes.c_int32 * max(n, 1))()
    blob = b''.join(blobs)
    _check(lib.blsgpu_deserialize(group, _ptr(blob), n, FMT_LEGACY if legacy else FMT_COMPRESSED, ctypes.cast(out, ctypes.c_void_p),
                                  ctypes.cast(st, ctypes.c_void_p)))
    raw = out.raw
    return [raw[osz * i:osz * (i + 1)] for i in range(n)], list(st)[:n]


def serialize(group, pts, fmt_in=FMT_RAW_PROJ, legacy=False):
    lib = init()
    n = len(pts)
    osz = 48 if group == 1 else 96
    out = ctypes.create_string_buffer(osz * max(n, 1))
    blob = b''.join(pts)
    _check(lib.blsgpu_serialize(group, _ptr(blob), n, fmt_in, FMT_LEGACY if legacy else FMT_COMPRESSED, ctypes.cast(out, ctypes.c_void_p), None))
    raw = out.raw
    return [raw[osz * i:osz * (i + 1)] for i in range(n)]


def sign_batch(sig_group, scheme, sks, msgs):
    """(pks, sigs) as RAW_PROJ byte strings for integer secret keys `sks` (sign side; builds test/bench inputs)."""
    lib = init()
    n = len(msgs)
    offs, blob = _offsets(msgs)
    pksz, sgsz = (288, 144) if sig_group == 1 else (144, 288)
    opk = ctypes.create_string_buffer(pksz * max(n, 1))
    osg = ctypes.create_string_buffer(sgsz * max(n, 1))
    skb = b''.join(int(s).to_bytes(32, 'little') for s in sks)
    _check(lib.blsgpu_sign_batch(sig_group, scheme, _ptr(skb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), n,
                                 ctypes.cast(opk, ctypes.c_void_p), ctypes.cast(osg, ctypes.c_void_p)))
    pk_raw, sg_raw = opk.raw, osg.raw          # .raw copies the whole buffer: take it once
    return ([pk_raw[pksz * i:pksz * (i + 1)] for i in range(n)], [sg_raw[sgsz * i:sgsz * (i + 1)] for i in range(n)])


def aggregate_partial(sig_group, scheme, pks, msgs, sig=None, fmt=FMT_RAW_PROJ):
    """(576-byte Fp12 record, first_bad) for a shard of (pk, msg) pairs; sig is folded in when given."""
    lib = init()
    offs, blob = _offsets(msgs)
    out = ctypes.create_string_buffer(576)
    fb = ctypes.c_int64(-2)
    pkb = b''.join(pks)
    _check(lib.blsgpu_aggregate_partial(sig_group, scheme, _ptr(pkb), _ptr(blob), ctypes.cast(offs, ctypes.c_void_p), len(pks),
                                        _ptr(sig) if sig is not None else None, fmt, ctypes.cast(out, ctypes.c_void_p), ctypes.byref(fb)))
    return out.raw, fb.value


def fp12_product_is_one(records):
    lib = init()
    r = ctypes.c_int32(-99)
    blob = b''.join(records)
    _check(lib.blsgpu_fp12_product_is_one(_ptr(blob), len(records), ctypes.byref(r)))
    return bool(r.value)


def signatures_from_tagged(sig_group, blobs):
    """Signature::try_from(&[u8]) per record (reference src/signature.rs:120-126): 49 / 97-byte serde_bare records ->
    (schemes, RAW_PROJ points, statuses).  A record of the wrong length is BAD_LENGTH without touching the device."""
    lib = init()
    width = 48 if sig_group == 1 else 96
    osz = 144 if sig_group == 1 else 288
    good = [i for i, b in enumerate(blobs) if len(b) == width + 1]
    n = len(good)
    tags = ctypes.create_string_buffer(max(n, 1))
    out = ctypes.create_string_buffer(osz * max(n, 1))
    st = (ctypes.c_int32 * max(n, 1))()
    blob = b''.join(blobs[i] for i in good)
    _check(lib.blsgpu_signatures_from_tagged(sig_group, _ptr(blob), n, ctypes.cast(tags, ctypes.c_void_p), ctypes.cast(out, ctypes.c_void_p),
                                             ctypes.cast(st, ctypes.c_void_p)))
    schemes, pts, sts = [None] * len(blobs), [None] * len(blobs), [BAD_LENGTH] * len(blobs)
    raw = out.raw
    for k, i in enumerate(good):
        schemes[i], pts[i], sts[i] = tags.raw[k], raw[osz * k:osz * (k + 1)], st[k]
    return schemes, pts, sts


def signatures_to_tagged(sig_group, schemes, sigs, fmt=FMT_RAW_PROJ):
    """Vec<u8>::from(&Signature<C>) per signature (reference src/signature.rs:112-118): scheme byte + compressed point."""
    lib = init()
    n = len(sigs)
    width = 48 if sig_group == 1 else 96
    out = ctypes.create_string_buffer((width + 1) * max(n, 1))
    _check(lib.blsgpu_signatures_to_tagged(sig_group, _ptr(bytes(schemes)), _ptr(b''.join(sigs)), n, fmt, ctypes.cast(out, ctypes.c_void_p)))
    raw = out.raw
    return [raw[(width + 1) * i:(width + 1) * (i + 1)] for i in range(n)]


def debug_wide_mul(a_list, b_list, reps=1):
    """[a * b^reps in Fp] through the row-wide multiplier; elements as 48-byte Montgomery words."""
    lib = init()
    n = len(a_list)
    out = ctypes.create_string_buffer(48 * max(n, 1))
    _check(lib.blsgpu_debug_wide_mul(_ptr(b''.join(a_list)), _ptr(b''.join(b_list)), n, reps, ctypes.cast(out, ctypes.c_void_p)))
    return [out.raw[48 * i:48 * (i + 1)] for i in range(n)]


_wide_defs = None


def wide_defs():
    """{name: value} of the engine's operation ids (WOP_*) and value arrays (WV_*), read from csrc/wide_tables.cuh"""
    global _wide_defs
    if _wide_defs is None:
        import re
        text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'wide_tables.cuh')).read()
        _wide_defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'^#define (W(?:OP|V)_\w+) (\d+)', text, re.M)}
    return _wide_defs


def debug_wide_program(steps, f_raw, reps=1):
    """steps: [(op, dst, a, b)] by name (e.g. ('MUL', 'T', 'F', 'U')); f_raw: twelve 48-byte Montgomery elements.  Returns
    the twelve elements of T after `reps` runs of the program on the row-wide engine."""
    lib = init()
    d = wide_defs()
    words = []
    def ref(r):          # 'T' or ('W', 3): an array or one value of it
        return d['WV_' + r] if isinstance(r, str) else d['WV_' + r[0]] + r[1]
    for op, dst, a, b in steps:
        words += [d['WOP_' + op] | ref(dst) << 16, ref(a) | ref(b) << 16]
    arr = (ctypes.c_uint32 * len(words))(*words)
    out = ctypes.create_string_buffer(576)
    _check(lib.blsgpu_debug_wide_program(ctypes.cast(arr, ctypes.c_void_p), len(steps), reps, _ptr(b''.join(f_raw)), ctypes.cast(out, ctypes.c_void_p)))
    return [out.raw[48 * i:48 * (i + 1)] for i in range(12)]


def first_duplicate_message(msgs):
    """(old, i) of the Basic scheme's duplicate-message rule (reference src/traits/sig_basic.rs:46-58), or None."""
    lib = init()
    offs, blob = _offsets(msgs)
    out = (ctypes.c_uint64 * 2)()
    _check(lib.blsgpu_first_duplicate_message(_ptr(blob), ctypes.cast(offs, ctypes.c_void_p), len(msgs), ctypes.cast(out, ctypes.c_void_p)))
    return None if out[1] == 2 ** 64 - 1 else (out[0], out[1])


# ------------------------------------------------------------------ tensor-level calls (device-resident shards)
class TensorOps:
    """The C ABI on torch uint8 / int32 / int64 tensors that live on this process's GPU: every call passes data_ptr()s,
    nothing is converted per item.  agora-blsful_amd/dist.py drives the one-process-per-GPU sharding through this object
    (tests/fake_backend.py offers the same methods on CPU tensors, backed by the oracle, for the gloo tests)."""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.lib = init(device.index if device.index is not None else -1)

    def _sync(self):
        """The library runs on its own (non-blocking) streams: tensors that torch kernels are still producing on torch's
        current stream must be complete before their pointers are handed over.  (The other direction needs nothing: every
        C-ABI call is blocking.)"""
        self.torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None

    def empty(self, n, dtype=None):
        return self.torch.empty(n, dtype=dtype or self.torch.uint8, device=self.device)

    def verify_batch(self, sg, scheme, pks, sigs, msgs, offs, n):
        self._sync()
        st = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_verify_batch(sg, scheme, self._p(pks), self._p(sigs), self._p(msgs), self._p(offs), n, FMT_RAW_PROJ, self._p(st)))
        return st[:n]

    def pop_verify_batch(self, sg, pks, proofs, n):
        self._sync()
        st = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_pop_verify_batch(sg, self._p(pks), self._p(proofs), n, FMT_RAW_PROJ, self._p(st)))
        return st[:n]

    def sig_proof_verify_batch(self, sg, scheme, us, vs, pks, ys, msgs, offs, n):
        self._sync()
        st = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_sig_proof_verify_batch(sg, scheme, self._p(us), self._p(vs), self._p(pks), self._p(ys), self._p(msgs), self._p(offs), n,
                                                      FMT_RAW_PROJ, self._p(st)))
        return st[:n]

    def signcrypt_valid_batch(self, sg, scheme, us, ws, vs, offs, n):
        self._sync()
        st = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_signcrypt_valid_batch(sg, scheme, self._p(us), self._p(ws), self._p(vs), self._p(offs), n, FMT_RAW_PROJ, self._p(st)))
        return st[:n]

    def point_sum(self, group, pts, n, scalars=None):
        self._sync()
        out = self.empty(144 if group == 1 else 288)
        if scalars is None:
            fn = self.lib.blsgpu_sum_g1 if group == 1 else self.lib.blsgpu_sum_g2
            _check(fn(self._p(pts), n, FMT_RAW_PROJ, self._p(out)))
        else:
            fn = self.lib.blsgpu_msm_g1 if group == 1 else self.lib.blsgpu_msm_g2
            _check(fn(self._p(pts), self._p(scalars), n, FMT_RAW_PROJ, self._p(out)))
        return out

    def multi_verify(self, sg, scheme, pks, n, sig, msg):
        self._sync()
        st = ctypes.c_int32(-99)
        _check(self.lib.blsgpu_multi_verify(sg, scheme, self._p(pks), n, self._p(sig), _ptr(msg), len(msg), FMT_RAW_PROJ, ctypes.byref(st)))
        return st.value

    def core_verify_one(self, sg, dst, pk, sig, msg):
        self._sync()
        st = ctypes.c_int32(-99)
        offs = (ctypes.c_uint64 * 2)(0, len(msg))
        _check(self.lib.blsgpu_core_verify(sg, _ptr(dst), len(dst), self._p(pk), self._p(sig), _ptr(msg), ctypes.cast(offs, ctypes.c_void_p), 1,
                                           FMT_RAW_PROJ, ctypes.byref(st)))
        return st.value

    def hash_to_point(self, sg, dst, msg):
        """H(msg) of the signature group of `sg` as a device tensor (RAW_PROJ).  Safe to call from a helper thread while
        another call of this object is in flight: the library leases a second context."""
        out = self.empty(144 if sg == 1 else 288)
        offs = (ctypes.c_uint64 * 2)(0, len(msg))
        fn = self.lib.blsgpu_hash_to_g1 if sg == 1 else self.lib.blsgpu_hash_to_g2
        _check(fn(_ptr(msg), ctypes.cast(offs, ctypes.c_void_p), 1, _ptr(dst), len(dst), self._p(out)))
        return out

    def core_verify_hashed_one(self, sg, pk, sig, hm):
        self._sync()
        st = ctypes.c_int32(-99)
        _check(self.lib.blsgpu_core_verify_hashed(sg, self._p(pk), self._p(sig), self._p(hm), 1, ctypes.byref(st)))
        return st.value

    def aggregate_partial(self, sg, scheme, pks, msgs, offs, n, sig=None):
        self._sync()
        """(576-byte record, first_bad) as DEVICE tensors: nothing crosses to the host."""
        rec, fb = self.empty(576), self.empty(1, self.torch.int64)
        _check(self.lib.blsgpu_aggregate_partial(sg, scheme, self._p(pks), self._p(msgs), self._p(offs), n, self._p(sig), FMT_RAW_PROJ,
                                                 self._p(rec), ctypes.cast(self._p(fb), ctypes.POINTER(ctypes.c_int64))))
        return rec, fb

    def fp12_product_is_one(self, recs, k):
        self._sync()
        r = ctypes.c_int32(-99)
        _check(self.lib.blsgpu_fp12_product_is_one(self._p(recs), k, ctypes.byref(r)))
        return bool(r.value)

    def first_duplicate(self, msgs, offs, n):
        self._sync()
        out = (ctypes.c_uint64 * 2)()
        _check(self.lib.blsgpu_first_duplicate_message(self._p(msgs), self._p(offs), n, ctypes.cast(out, ctypes.c_void_p)))
        return None if out[1] == 2 ** 64 - 1 else (out[0], out[1])

    def serialize(self, group, pts, n, legacy=False):
        self._sync()
        out = self.empty((48 if group == 1 else 96) * max(n, 1))
        _check(self.lib.blsgpu_serialize(group, self._p(pts), n, FMT_RAW_PROJ, FMT_LEGACY if legacy else FMT_COMPRESSED, self._p(out), None))
        return out[:(48 if group == 1 else 96) * n]

    def sort_keys(self, kb, n, width):
        self._sync()
        perm = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_sort_keys(self._p(kb), n, width, self._p(perm)))
        return perm[:n]

    def keys_digest(self, kb, perm, n, width):
        self._sync()
        out = self.empty(32)
        _check(self.lib.blsgpu_sorted_keys_digest(self._p(kb), self._p(perm), n, width, self._p(out)))
        return out

    def coefficients_for_range(self, digest, perm, n, base, count):
        self._sync()
        scal = self.empty(32 * max(count, 1))
        st = ctypes.c_int32(-99)
        _check(self.lib.blsgpu_coefficients_for_range(self._p(digest), self._p(perm), n, base, count, self._p(scal), ctypes.byref(st)))
        return scal[:32 * count], st.value

    def first_occurrence(self, kb, perm, n, width):
        self._sync()
        out = self.empty(max(n, 1), self.torch.int32)
        _check(self.lib.blsgpu_first_occurrence(self._p(kb), self._p(perm), n, width, self._p(out)))
        return out[:n]

    def is_identity(self, group, pt):
        self._sync()
        return int(self.serialize(group, pt, 1)[0].item()) == 0xc0


def profile_enable(on=True):
    _check(init().blsgpu_profile_enable(1 if on else 0))


def profile_read():
    """{kernel name: (total device ms, launches)} accumulated since profile_enable()."""
    lib = init()
    out = {}
    for k in range(lib.blsgpu_profile_count()):
        name = ctypes.create_string_buffer(64)
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        _check(lib.blsgpu_profile_get(k, name, 64, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            out[name.value.decode()] = (ms.value, cnt.value)
    return out


# ------------------------------------------------------------------ reference-shaped wrapper types
class Impl:
    def __init__(self, sig_group):
        self.sig_group = sig_group
        self.name = 'Bls12381G%dImpl' % sig_group


Bls12381G1Impl = Impl(1)
Bls12381G2Impl = Impl(2)


class PublicKey:
    """PublicKey<C>(pub C::PublicKey): reference src/public_key.rs:5-11.  `raw` = RAW_PROJ bytes."""

    def __init__(self, impl, raw):
        self.impl, self.raw = impl, bytes(raw)


class Signature:
    """Signature<C> {Basic, MessageAugmentation, ProofOfPossession}: reference src/signature.rs:25-44."""

    def __init__(self, impl, scheme, raw):
        self.impl, self.scheme, self.raw = impl, scheme, bytes(raw)

    def verify(self, pk, msg):
        """reference src/signature.rs:130-138; raises BlsError on failure, returns None on Ok(())."""
        st = verify_batch(self.impl.sig_group, self.scheme, [pk.raw], [self.raw], [bytes(msg)])[0]
        e = error_from_status(st)
        if e:
            raise e

    def verify_secure(self, public_keys, msg):
        """reference src/signature.rs:177-197."""
        st = verify_secure(self.impl.sig_group, self.scheme, [p.raw for p in public_keys], self.raw, bytes(msg), MODERN)
        e = error_from_status(st)
        if e:
            raise e

    def verify_secure_with_mode(self, public_keys, msg, fmt):
        """reference src/signature.rs:256-276."""
        st = verify_secure(self.impl.sig_group, self.scheme, [p.raw for p in public_keys], self.raw, bytes(msg), fmt)
        e = error_from_status(st)
        if e:
            raise e


class MultiPublicKey:
    """reference src/multi_public_key.rs:5-11; built lazily: the sum runs on the GPU inside MultiSignature.verify."""

    def __init__(self, impl, keys):
        self.impl, self.keys = impl, list(keys)

    @staticmethod
    def from_public_keys(keys):
        return MultiPublicKey(keys[0].impl, keys)


class MultiSignature:
    """reference src/multi_signature.rs:6-25."""

    def __init__(self, impl, scheme, raw):
        self.impl, self.scheme, self.raw = impl, scheme, bytes(raw)

    def verify(self, mpk, msg):
        """reference src/multi_signature.rs:127-135."""
        st = multi_verify(self.impl.sig_group, self.scheme, [k.raw for k in mpk.keys], self.raw, bytes(msg))
        e = error_from_status(st)
        if e:
            raise e


class AggregateSignature:
    """reference src/aggregate_signature.rs:33-52."""

    def __init__(self, impl, scheme, raw):
        self.impl, self.scheme, self.raw = impl, scheme, bytes(raw)

    def verify(self, data):
        """data: [(PublicKey, msg)]; reference src/aggregate_signature.rs:230-239."""
        st, aux = aggregate_verify(self.impl.sig_group, self.scheme, [p.raw for p, _ in data], [bytes(m) for _, m in data], self.raw)
        e = error_from_status(st, aux, aggregate=True)
        if e:
            raise e
