"""world_size-2 tests of the sharded verify path (agora-blsful_amd/dist.py) over gloo: on the CPU with the
oracle-backed fake backend, and on the GPU box with two processes sharing the card (real kernels, gloo exchange)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import util

EXPECT_OK = 0


def run_world(backend, tmp_path, world=2):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = str(s.getsockname()[1])
    s.close()
    outs = [str(tmp_path / ('r%d.json' % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(util.ROOT, 'tests', 'dist_worker.py'), str(r), str(world), port, backend, outs[r]])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    res = [json.load(open(o)) for o in outs]
    assert all(r == res[0] for r in res), 'ranks disagree'
    return res[0]


def check(res):
    for sg in (1, 2):
        assert res['verify_batch_%d' % sg] == [0, 0, 0, 0, 0, 1, 0]
        assert res['multi_ok_%d' % sg] == 0 and res['multi_bad_%d' % sg] == 1
        for scheme in (0, 1):
            assert res['agg_ok_%d_%d' % (sg, scheme)] == [0, [0, 0]]
            assert res['agg_bad_%d_%d' % (sg, scheme)] == [1, [0, 0]]
            # Basic: duplicates win over everything; Aug: no duplicate rule, the mutated message just fails
            assert res['agg_dup_%d_%d' % (sg, scheme)] == ([4, [1, 6]] if scheme == 0 else [1, [0, 0]])
            assert res['agg_pkid_%d_%d' % (sg, scheme)] == [3, [5, 0]]          # first identity key, 1-based
            assert res['agg_sigid_%d_%d' % (sg, scheme)] == [2, [0, 0]]         # signature identity before key identity
        for mode in ([0] if sg == 1 else [0, 1]):
            assert res['secure_ok_%d_%d' % (sg, mode)] == 0
            assert res['secure_sub_%d_%d' % (sg, mode)] == 1
        assert res['secure_empty_%d' % sg] == [0, 1]
        assert res['pop_%d' % sg] == [0, 0, 1, 0, 0, 0, 0]
        assert res['proof_%d' % sg] == [0, 1, 0, 0, 0, 0, 9]
        assert res['signcrypt_%d' % sg] == [True, True, True, True, False, True, True]
        for mode in ([0] if sg == 1 else [0, 1]):
            assert res['aggsec_%d_%d' % (sg, mode)] == [0, True]
        assert res['agg_one_%d' % sg] == [0, [0, 0]]                            # one rank's shard is empty


def test_shard_range(pkg):
    from agora_blsful_amd import dist as bd
    for n in (0, 1, 7, 64, 65536, 262144 + 3):
        for w in (1, 2, 3, 8):
            r = [bd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_sharded_paths_gloo_world2_cpu(pkg, tmp_path):
    check(run_world('fake', tmp_path))


@pytest.mark.gpu
def test_sharded_paths_gloo_world2_gpu(api, tmp_path):
    check(run_world('hip', tmp_path))
