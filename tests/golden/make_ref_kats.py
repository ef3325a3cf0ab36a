"""Extract the known-answer DATA held by the reference's own tests into tests/golden/ref_kats.json.

Only byte strings (keys, signatures, messages) are extracted -- no reference source text is kept.
Sources (read as text; /root/reference exists only in the build container):
  tests/cpp_integration_test.rs:19-82,196-204   C++ (relic/bls-signatures) vectors: 3 sk, 3 pk, 3 sig, naive agg
  tests/secure_aggregation_test.rs:146-208      57-signer production vector (sig, 57 pks, msg)
Run:  python tests/golden/make_ref_kats.py
"""
import json
import os
import re

REF = '/root/reference/tests'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_kats.json')


def byte_arrays(text):
    out = {}
    for m in re.finditer(r'const (\w+): \[u8; \d+\] = \[(.*?)\];', text, re.S):
        out[m.group(1)] = bytes(int(x, 16) for x in re.findall(r'0x([0-9a-fA-F]{2})', m.group(2))).hex()
    return out


def main():
    cpp = open(os.path.join(REF, 'cpp_integration_test.rs')).read()
    arrs = byte_arrays(cpp)
    m = re.search(r'let normal_agg_bytes = \[(.*?)\];', cpp, re.S)
    naive = bytes(int(x, 16) for x in re.findall(r'0x([0-9a-fA-F]{2})', m.group(1))).hex()
    sec = open(os.path.join(REF, 'secure_aggregation_test.rs')).read()
    body = sec[sec.index('fn test_large_scale_aggregate_signature_verification'):]
    sig_hex = re.search(r'let sig_hex = "([0-9a-f]+)"', body).group(1)
    keys = re.findall(r'^\s*"([0-9a-f]{96})",?\s*$', body, re.M)
    msg_hex = re.search(r'let message_hex = "([0-9a-f]+)"', body).group(1)
    assert len(keys) == 57 and len(sig_hex) == 192
    doc = {
        '_source': 'dashpay/agora-blsful tests/cpp_integration_test.rs:19-82,196-204; '
                   'tests/secure_aggregation_test.rs:146-208 (data only)',
        'cpp': {
            'impl': 'G2', 'scheme': 'Basic', 'message': arrs['MESSAGE_HELLO'],
            'sk': [arrs['CPP_SK%d_BYTES' % i] for i in (1, 2, 3)],
            'pk': [arrs['CPP_PK%d_BYTES' % i] for i in (1, 2, 3)],
            'sig': [arrs['CPP_SIG%d_BYTES' % i] for i in (1, 2, 3)],
            'naive_agg_sig_pk12': naive,
            'expect': {'sk_to_pk': True, 'sig_verify': True, 'naive_agg_verify_secure_pk12': False},
        },
        'prod57': {
            'impl': 'G2', 'scheme': 'Basic', 'format': 'Modern',
            'sig': sig_hex, 'pks': keys, 'message': msg_hex,
            'expect': {'verify_secure': True},
        },
    }
    with open(OUT, 'w') as f:
        json.dump(doc, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
