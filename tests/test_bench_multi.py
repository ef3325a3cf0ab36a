"""bench.py's own N > 1 control flow (not just agora-blsful_amd/dist.py): two ranks sharing the card over gloo run the headline
and every other_configs entry; a deliberately failing entry leaves every rank exiting non-zero instead of hanging
(VERDICT r2 "next round" item 2)."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

import util


def launch(extra, timeout):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = str(s.getsockname()[1])
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', port,
           os.path.join(util.ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--items', '1024', '--size', '2048', '--steps', '1', '--warmup', '1'] + extra
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=util.ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r.returncode, (json.loads(lines[-1]) if lines else None), time.time() - t0, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_two_ranks_gloo_runs_every_config(api):
    rc, out, _, err = launch([], 900)
    assert rc == 0 and out is not None, err
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['scaling'] == 'weak' and out['collective_ms_per_step'] == 0.0
    oc = out['other_configs']
    for name in ('config3_multi_verify_1048576', 'config4_aggregate_verify_262144', 'config5_verify_secure_65536_g1impl_modern',
                 'config5_verify_secure_65536_g2impl_modern', 'config5_verify_secure_65536_g2impl_legacy', 'config2_g2impl_65536'):
        assert 'error' not in oc[name], (name, oc[name])
        assert oc[name]['rccl_ranks'] == 2 and oc[name]['n_gpus'] == 2 and oc[name]['value'] > 0
        assert 'collective_ms_per_step' in oc[name]
    assert oc['config4_aggregate_verify_262144']['collective_ms_per_step'] > 0       # the Fp12 records did cross the process group


@pytest.mark.gpu
def test_bench_failing_extra_exits_nonzero_on_every_rank(api):
    rc, out, dt, err = launch(['--fail-extra', 'config4_aggregate_verify_262144:1', '--pg-timeout', '30'], 600)
    assert rc != 0, 'the launcher must report the failure'
    assert dt < 400, 'ranks must leave within the process-group timeout, not hang'
    assert out is not None and out['value'] > 0, 'the headline line survives a failing extra'
    oc = out['other_configs']
    assert 'error' not in oc['config3_multi_verify_1048576']
    assert 'deliberate failure' in oc['config4_aggregate_verify_262144']['error']
    assert 'config5_verify_secure_65536_g1impl_modern' not in oc                      # the ride-along ends at the failure


@pytest.mark.gpu
def test_bench_hanging_extra_still_prints_the_headline(api):
    """one rank never reaches an entry's collectives: the watchdog over the ride-along prints the headline line (with the entries that
    finished) and every rank leaves -- before the process group's own timeout can take the process down without a line"""
    rc, out, dt, err = launch(['--fail-extra', 'config4_aggregate_verify_262144:1:hang', '--pg-timeout', '300', '--extras-timeout', '45'], 600)
    assert rc != 0 and dt < 300
    assert out is not None and out['value'] > 0
    oc = out['other_configs']
    assert 'error' not in oc['config3_multi_verify_1048576']
    assert 'timeout' in oc['config4_aggregate_verify_262144']['error']
