"""bench.py's result-line arithmetic on a synthetic kernel profile (CPU): kernel_ms is per STEP with a launch count, the roofline is
per LAUNCH of the kernel with the largest per-step total on the main stream, and a chunked kernel's launch covers one chunk
(config 4's line divided the whole call's bytes by one chunk's launch -- 4x too high)."""
import importlib.util
import os

import util


def load_bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(util.ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # top level imports no torch
    return m


def test_kernel_ms_is_per_step_and_roofline_per_launch():
    b = load_bench()
    steps = 3
    # config 4 as profiled in round 3, times per step x 3 steps: k_linesp 4 launches of 2.17 ms, k_line_quad 4 x 1.78, fold 11 x 0.52, prepare 1 x 5.4;
    # the tail stream's launches run beside them and must never be "dominant" however long they are
    prof = {'k_linesp': (3 * 4 * 2.17, 3 * 4), 'k_line_quad': (3 * 4 * 1.78, 3 * 4), 'k_f12_fold4': (3 * 11 * 0.52, 3 * 11), 'k_prepare_agg': (3 * 5.4, 3),
            'tail_stream_overlapped': (3 * 20.0, 3 * 2), 'k_unused': (0.0, 0)}
    km = b.kernel_ms(prof, steps)
    assert km['k_linesp'] == [8.68, 4.0] and km['k_prepare_agg'] == [5.4, 1.0] and 'k_unused' not in km
    rl = b.roofline_of(prof, steps, 320, 262144)
    assert rl['kernel'] == 'k_linesp' and rl['launches_per_step'] == 4.0 and rl['units_per_launch'] == 65536.0
    assert abs(rl['avg_launch_ms'] - 2.17) < 1e-6
    assert abs(rl['achieved'] - 320 * 65536 / 2.17e-3 / 1e9) < 1e-3          # 9.7 GB/s, not 38.6
    assert abs(rl['frac'] - rl['achieved'] / 8000.0) < 1e-7 and rl['algorithmic_bytes_per_launch'] == 320 * 65536
    # a forced kernel (config 3 reports the kernel that reads the keys)
    rl = b.roofline_of(prof, steps, 288, 1048576, 'k_prepare_agg')
    assert rl['kernel'] == 'k_prepare_agg' and rl['units_per_launch'] == 1048576.0


def test_self_launch_is_chosen_only_without_a_launcher(monkeypatch):
    """--gpus N > 1 starts its own ranks only when neither RANK nor WORLD_SIZE is set; --gpus 1 never does"""
    b = load_bench()
    calls = []
    monkeypatch.setattr(b, 'self_launch', lambda args, argv: calls.append(list(argv)) or 0)

    class Stop(Exception):
        pass

    def harness(args):
        raise Stop()
    monkeypatch.setattr(b, 'Harness', harness)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    try:
        b.main(['--gpus', '4', '--steps', '1'])
    except SystemExit as e:
        assert e.code == 0
    assert calls == [['--gpus', '4', '--steps', '1']]
    calls.clear()
    for argv, env in ((['--gpus', '1'], {}), (['--gpus', '4'], {'RANK': '0', 'WORLD_SIZE': '4'})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            b.main(argv)
        except Stop:
            pass
        assert calls == []
