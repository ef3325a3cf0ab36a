"""The operation tables and programs of the row-wide engine (csrc/wide_tables.cuh, the single-verification latency path):
tools/gen_wide_tables.py derives them from the tower / Miller-step formulas and checks them against the oracle on plain
integers -- every Fp12 operation, the hard part of the final exponentiation, WHOLE pairing checks (line coefficients,
Miller loop, final exponentiation; a valid and an invalid signature), and the sixteen-point sum programs of both groups
(complete projective additions; repeated, opposite and identity inputs; Jacobian and homogeneous ends) -- before writing the header.  This test re-runs those
checks and makes sure the committed header is what the generator produces."""
import importlib.util
import os

import util


def test_engine_tables_match_the_oracle_and_the_committed_header(tmp_path):
    spec = importlib.util.spec_from_file_location('gen_wide_tables', os.path.join(util.ROOT, 'tools', 'gen_wide_tables.py'))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert g.self_check()
    g.check_programs()
    g.check_point_programs()
    out = tmp_path / 'wide_tables.cuh'
    g.emit(str(out))
    csrc = os.path.join(util.ROOT, 'agora-blsful_amd', 'csrc')
    assert open(out).read() == open(os.path.join(csrc, 'wide_tables.cuh')).read()
    assert open(tmp_path / 'wide_rows.cuh').read() == open(os.path.join(csrc, 'wide_rows.cuh')).read()
