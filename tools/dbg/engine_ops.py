"""per-operation time of the row-wide engine (blsgpu_debug_wide_program): programs of one repeated operation on one workgroup"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
import util
pkg = ge.import_pkg(); api = pkg.api; api.init()
f = [util.fp_raw(1234567 + 99 * i) for i in range(12)]
N, REPS = 200, 10
api.debug_wide_program([('COPY', 'T', 'F', 'F')], f)
for op in ('COPY', 'CONJ', 'CYC_SQR', 'SQR', 'MUL', 'MUL_LINE', 'FROB1', 'FPINV'):
    api.profile_enable(True)
    api.debug_wide_program([(op, 'T', 'F', 'U') if op != 'FPINV' else (op, ('T', 0), ('F', 0), ('F', 0))] * N, f, REPS)
    ms = api.profile_read()['k_wide'][0]
    api.profile_enable(False)
    print('%-9s %.3f us per step' % (op, ms * 1e3 / (N * REPS)), flush=True)
