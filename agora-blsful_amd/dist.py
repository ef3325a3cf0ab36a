"""One-process-per-GPU sharding of the verify path.  Every function takes the caller's LOCAL shard
(contiguous item range [base, base + n_local) of the global input) and returns the GLOBAL result on every rank, with
the reference's semantics and error precedence.

  verify_batch       independent items: no exchange on the data path, only the verdicts are gathered
  multi_verify       local point sum -> all-gather of one group element per rank (144/288 B) -> fold -> one verify
  aggregate_verify   local Miller products -> all-gather of one Fp12 record per rank (576 B) + first-bad indices ->
                     fold + final exponentiation; Basic's duplicate-message rule is decided on gathered SHA-256 digests
  verify_secure      local compress -> all-gather of the serialised keys (n * 48/96 B) -> every rank derives the same
                     sort / H / t_i -> local MSM over its own keys with their coefficients -> all-gather of one group
                     element per rank -> fold -> one core_verify
  pop_verify_batch   independent (key, proof) items like verify_batch
  aggregate_secure   the sign-side twin of verify_secure: same gathered keys and coefficients, local MSM over the local
                     SIGNATURES, all-gather of one group element per rank -> fold

`backend` is the C-ABI wrapper module (agora-blsful_amd/api.py); `pg` is torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; the payloads are tiny so the exchange is latency-bound) or None for a single process.  RCCL has no
user-defined reduction and neither Fp12 products nor point additions are element-wise, hence all-gather + local fold.
"""
import hashlib

OK, INVALID_SIGNATURE, SIG_IDENTITY, PK_IDENTITY, DUPLICATE_MESSAGE, INVALID_COEFFICIENT = 0, 1, 2, 3, 4, 5
BASIC, AUG, POP = 0, 1, 2
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def shard_range(n, rank, world):
    """Contiguous, balanced, order-preserving partition of range(n)."""
    return n * rank // world, n * (rank + 1) // world


class Sharded:
    def __init__(self, backend, pg=None, device=None):
        self.be, self.pg = backend, pg
        self.rank = pg.get_rank() if pg is not None else 0
        self.world = pg.get_world_size() if pg is not None else 1
        self.device = device

    # ---- collectives on byte strings
    def _gather_objects(self, obj):
        if self.pg is None:
            return [obj]
        out = [None] * self.world
        self.pg.all_gather_object(out, obj)
        return out

    def _gather_fixed(self, b):
        """all-gather of one fixed-size record per rank as a uint8 tensor (the data-path exchange)."""
        if self.pg is None:
            return [bytes(b)]
        import torch
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        if self.device is not None:
            t = t.to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.pg.all_gather(outs, t)
        return [bytes(o.cpu().numpy().tobytes()) for o in outs]

    # ---- config 2
    def verify_batch(self, sig_group, scheme, pks, sigs, msgs):
        local = self.be.verify_batch(sig_group, scheme, pks, sigs, msgs)
        return [s for part in self._gather_objects(local) for s in part]

    # ---- config 3
    def multi_verify(self, sig_group, scheme, pks, sig, msg):
        pk_group = 2 if sig_group == 1 else 1
        partial = self.be.point_sum(pk_group, pks)
        apk = self.be.point_sum(pk_group, self._gather_fixed(partial))
        return self.be.verify_batch(sig_group, scheme, [apk], [sig], [msg])[0]

    # ---- config 4
    def aggregate_verify(self, sig_group, scheme, pks, msgs, sig, base):
        """returns (status, (aux0, aux1)) like blsgpu_aggregate_verify, with global indices."""
        rec, fb = self.be.aggregate_partial(sig_group, scheme, pks, msgs, sig if self.rank == 0 else None)
        if scheme == BASIC:
            digests = [d for part in self._gather_objects([hashlib.sha256(m).digest() for m in msgs]) for d in part]
            seen = {}
            for i, d in enumerate(digests):          # reference src/traits/sig_basic.rs:46-58
                if d in seen:
                    return DUPLICATE_MESSAGE, (seen[d], i)
                seen[d] = i
        info = self._gather_objects((base, len(pks), fb))
        if info[0][2] == info[0][1]:                 # rank 0 saw the identity signature (local index n)
            return SIG_IDENTITY, (0, 0)
        firsts = [b + f for (b, n, f) in info if 0 <= f < n]
        if firsts:
            return PK_IDENTITY, (min(firsts) + 1, 0)  # 1-based, reference src/traits/sig_core.rs:163-166
        ok = self.be.fp12_product_is_one(self._gather_fixed(rec))
        return (OK if ok else INVALID_SIGNATURE), (0, 0)

    # ---- config 5
    def verify_secure(self, sig_group, scheme, pks, sig, msg, base, ser_format=0):
        pk_group = 2 if sig_group == 1 else 1
        local_bytes = self.be.serialize(pk_group, pks, legacy=bool(ser_format)) if pks else []
        parts = self._gather_objects((base, local_bytes))
        parts.sort(key=lambda p: p[0])
        all_bytes = [b for (_, bs) in parts for b in bs]
        if not all_bytes:                            # reference src/secure_aggregation.rs:189-195
            ident = self.be.serialize(sig_group, [sig])[0][0] == 0xc0
            return OK if ident else INVALID_SIGNATURE
        st, perm, ts = self.be.secure_coefficients(all_bytes)
        if st != OK:
            return st
        pos = {orig: p for p, orig in enumerate(perm)}
        scal = [ts[pos[base + i]] for i in range(len(pks))]
        partial = self.be.point_sum(pk_group, pks, scal)
        apk = self.be.point_sum(pk_group, self._gather_fixed(partial))
        return self.be.core_verify(sig_group, self.be.DST[(sig_group, scheme)], [apk], [sig], [msg])[0]


    # ---- N3: ProofOfPossession::verify, independent items (reference src/proof_of_possession.rs:79-81)
    def pop_verify_batch(self, sig_group, pks, proofs):
        local = self.be.pop_verify_batch(sig_group, pks, proofs)
        return [s for part in self._gather_objects(local) for s in part]

    # ---- N4: the other per-item two-pairing checks (independent items: shard, no data-path collective)
    def sig_proof_verify_batch(self, sig_group, scheme, commitments, proofs, pks, ys, msgs):
        """ProofOfKnowledge::verify per item (reference src/traits/sig_proof.rs:102-142): statuses in global order."""
        local = self.be.sig_proof_verify_batch(sig_group, scheme, commitments, proofs, pks, ys, msgs)
        return [s for part in self._gather_objects(local) for s in part]

    def signcrypt_valid_batch(self, sig_group, scheme, us, ws, vs):
        """SignCryptCiphertext::is_valid per item (reference src/traits/sign_crypt.rs:69-77): bools in global order."""
        local = self.be.signcrypt_valid_batch(sig_group, scheme, us, ws, vs)
        return [s for part in self._gather_objects(local) for s in part]

    # ---- N1: aggregate_secure[_with_mode] (reference src/secure_aggregation.rs:110-169,338-352)
    def aggregate_secure(self, sig_group, pks, sigs, base, ser_format=0):
        """(status, RAW_PROJ aggregate signature) on every rank.  The reference looks every sorted key up with `position`,
        so duplicated keys all take the signature of their FIRST occurrence: that owner adds up their coefficients."""
        pk_group = 2 if sig_group == 1 else 1
        local_bytes = self.be.serialize(pk_group, pks, legacy=bool(ser_format)) if pks else []
        parts = self._gather_objects((base, local_bytes))
        parts.sort(key=lambda p: p[0])
        all_bytes = [b for (_, bs) in parts for b in bs]
        if not all_bytes:
            return OK, self.be.point_sum(sig_group, [])
        st, perm, ts = self.be.secure_coefficients(all_bytes)
        if st != OK:
            return st, None
        first = {}
        for g, b in enumerate(all_bytes):
            first.setdefault(b, g)
        coef = {}
        for p, orig in enumerate(perm):
            owner = first[all_bytes[orig]]
            coef[owner] = (coef.get(owner, 0) + ts[p]) % R_ORDER
        mine = [(sigs[i], coef[base + i]) for i in range(len(pks)) if coef.get(base + i)]
        partial = self.be.point_sum(sig_group, [m[0] for m in mine], [m[1] for m in mine])
        return OK, self.be.point_sum(sig_group, self._gather_fixed(partial))
