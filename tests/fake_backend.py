"""TEST-ONLY stand-in for agora-blsful_amd/api.py backed by the oracle, so that the sharding/orchestration logic of
agora-blsful_amd/dist.py can be exercised with world_size 2 over gloo on a machine without a GPU.  The product never
imports this.  FakeOps offers the methods of api.TensorOps on CPU tensors: dist.py runs the SAME code path on it that
it runs on device tensors."""
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


# ------------------------------------------------------------------ the tensor-level interface of api.TensorOps
class FakeOps:
    def __init__(self):
        import torch
        self.torch, self.device = torch, torch.device('cpu')

    def empty(self, n, dtype=None):
        return self.torch.zeros(n, dtype=dtype or self.torch.uint8)

    def _t(self, b):
        return self.torch.frombuffer(bytearray(b), dtype=self.torch.uint8) if len(b) else self.torch.zeros(0, dtype=self.torch.uint8)

    @staticmethod
    def _rows(t, n, sz):
        b = bytes(t.numpy().tobytes())
        return [b[sz * i:sz * (i + 1)] for i in range(n)]

    @staticmethod
    def _msgs(blob, offs, n):
        b, o = bytes(blob.numpy().tobytes()), [int(x) for x in offs[:n + 1].tolist()]
        return [b[o[i]:o[i + 1]] for i in range(n)]

    def _st(self, lst):
        return self.torch.tensor(lst, dtype=self.torch.int32)

    @staticmethod
    def _sizes(sg):
        return (288, 144) if sg == 1 else (144, 288)

    def verify_batch(self, sg, scheme, pks, sigs, msgs, offs, n):
        ps, ss = self._sizes(sg)
        return self._st(verify_batch(sg, scheme, self._rows(pks, n, ps), self._rows(sigs, n, ss), self._msgs(msgs, offs, n)))

    def pop_verify_batch(self, sg, pks, proofs, n):
        ps, ss = self._sizes(sg)
        return self._st(pop_verify_batch(sg, self._rows(pks, n, ps), self._rows(proofs, n, ss)))

    def sig_proof_verify_batch(self, sg, scheme, us, vs, pks, ys, msgs, offs, n):
        ps, ss = self._sizes(sg)
        yl = [int.from_bytes(y, 'little') for y in self._rows(ys, n, 32)]
        return self._st(sig_proof_verify_batch(sg, scheme, self._rows(us, n, ss), self._rows(vs, n, ss), self._rows(pks, n, ps), yl,
                                               self._msgs(msgs, offs, n)))

    def signcrypt_valid_batch(self, sg, scheme, us, ws, vs, offs, n):
        ps, ss = self._sizes(sg)
        ok = signcrypt_valid_batch(sg, scheme, self._rows(us, n, ps), self._rows(ws, n, ss), self._msgs(vs, offs, n))
        return self._st([0 if x else 1 for x in ok])

    def point_sum(self, group, pts, n, scalars=None):
        sz = 144 if group == 1 else 288
        sc = None if scalars is None else [int.from_bytes(x, 'little') for x in self._rows(scalars, n, 32)]
        return self._t(point_sum(group, self._rows(pts, n, sz), sc))

    def multi_verify(self, sg, scheme, pks, n, sig, msg):
        C = _impl(sg)
        ps, ss = self._sizes(sg)
        P = [_pk(sg, b) for b in self._rows(pks, n, ps)]
        return _status(lambda: ref.multi_sig_verify(C, scheme, P, _sig(sg, bytes(sig.numpy().tobytes())), msg))

    def core_verify_one(self, sg, dst, pk, sig, msg):
        return core_verify(sg, dst, [bytes(pk.numpy().tobytes())], [bytes(sig.numpy().tobytes())], [msg])[0]

    def hash_to_point(self, sg, dst, msg):
        h = _impl(sg).hash_to_point(msg, dst)
        return self._t((util.g1_raw if sg == 1 else util.g2_raw)(h))

    def core_verify_hashed_one(self, sg, pk, sig, hm):
        C = _impl(sg)
        P, S = _pk(sg, bytes(pk.numpy().tobytes())), _sig(sg, bytes(sig.numpy().tobytes()))
        Hm = _sig(sg, bytes(hm.numpy().tobytes()))
        if S is None:                                   # reference src/traits/sig_core.rs:126-135
            return SIG_IDENTITY
        if P is None:
            return PK_IDENTITY
        neg = C.pk_curve.neg(C.pk_gen)
        pairs = [(Hm, P), (S, neg)] if sg == 1 else [(P, Hm), (neg, S)]
        return OK if c.final_exponentiation(c.miller_loop(pairs)) == c.F12_ONE else INVALID_SIGNATURE

    def aggregate_partial(self, sg, scheme, pks, msgs, offs, n, sig=None):
        ps, _ = self._sizes(sg)
        rec, fb = aggregate_partial(sg, scheme, self._rows(pks, n, ps), self._msgs(msgs, offs, n),
                                    None if sig is None else bytes(sig.numpy().tobytes()))
        return self._t(rec), self.torch.tensor([fb], dtype=self.torch.int64)

    def fp12_product_is_one(self, recs, k):
        return fp12_product_is_one(self._rows(recs, k, 576))

    def first_duplicate(self, msgs, offs, n):
        seen = {}
        for i, m in enumerate(self._msgs(msgs, offs, n)):      # reference src/traits/sig_basic.rs:46-58
            if m in seen:
                return seen[m], i
            seen[m] = i
        return None

    def serialize(self, group, pts, n, legacy=False):
        sz = 144 if group == 1 else 288
        return self._t(b''.join(serialize(group, self._rows(pts, n, sz), legacy)))

    def sort_keys(self, kb, n, width):
        keys = self._rows(kb, n, width)
        return self.torch.tensor(sorted(range(n), key=lambda i: keys[i]), dtype=self.torch.int32)   # sorted() is stable

    def keys_digest(self, kb, perm, n, width):
        import hashlib
        keys = self._rows(kb, n, width)
        return self._t(hashlib.sha256(b''.join(keys[int(i)] for i in perm.tolist())).digest())

    def coefficients_for_range(self, digest, perm, n, base, count):
        import hashlib
        H = bytes(digest.numpy().tobytes())
        out, st = [b'\0' * 32] * count, 0
        for p, g in enumerate(perm.tolist()):
            if base <= g < base + count:
                t = int.from_bytes(hashlib.sha256(p.to_bytes(4, 'big') + H).digest(), 'big') % c.R
                if t == 0:
                    st = 5
                out[g - base] = t.to_bytes(32, 'little')
        return self._t(b''.join(out)), st

    def first_occurrence(self, kb, perm, n, width):
        keys = self._rows(kb, n, width)
        first = {}
        for i, k in enumerate(keys):
            first.setdefault(k, i)
        return self.torch.tensor([first[keys[int(g)]] for g in perm.tolist()], dtype=self.torch.int32)

    def is_identity(self, group, pt):
        return (g1_from_raw if group == 1 else g2_from_raw)(bytes(pt.numpy().tobytes())) is None
