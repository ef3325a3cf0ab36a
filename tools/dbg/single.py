"""One single-item verify_batch call repeated a few times (for rocprofv3 --pmc / --kernel-trace of the latency path)."""
import sys, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; api.init()
msg = hashlib.sha256(b'single').digest()
pks, sigs = api.sign_batch(1, api.POP, [0x1234567], [msg])
for _ in range(4):
    assert api.verify_batch(1, api.POP, pks, sigs, [msg]) == [0]
