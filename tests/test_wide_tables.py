"""The operation tables and programs of the row-wide engine (csrc/wide_tables.cuh: the single-verdict paths and the point-sum
tails): tools/gen_wide_tables.py derives them from the tower / Miller-step / addition-law formulas; tests/wide_tables_check.py
simulates the engine's semantics on plain integers and compares with the oracle -- every Fp12 operation, the hard part of the
final exponentiation, WHOLE pairing checks (line coefficients, Miller loop, final exponentiation; a valid and an invalid
signature) uncut and cut where the inputs become known, the sixteen-point sum programs of both groups (complete projective
additions; repeated, opposite and identity inputs; Jacobian and homogeneous ends) and the cofactor clearing of hash-to-G2.  This
test runs those checks and makes sure the committed header is what the generator writes."""
import os

import util
import wide_tables_check as chk


def test_engine_tables_match_the_oracle_and_the_committed_header(tmp_path):
    assert chk.self_check()
    chk.check_programs()
    chk.check_point_programs()
    out = tmp_path / 'wide_tables.cuh'
    chk.g.emit(str(out))
    csrc = os.path.join(util.ROOT, 'agora-blsful_amd', 'csrc')
    assert open(out).read() == open(os.path.join(csrc, 'wide_tables.cuh')).read()
    assert open(tmp_path / 'wide_rows.cuh').read() == open(os.path.join(csrc, 'wide_rows.cuh')).read()
