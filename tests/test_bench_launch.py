"""bench.py host logic that does not need a GPU: the self-launching `--gpus N` form (`python3 bench.py
--gpus 8` from a bare shell used to die on an assert) and the per-mode roofline block (bf16 runs used to print
fractions of 4.2-4.8 against the fp32 peak)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _bare_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID')}
    env['OMP_NUM_THREADS'] = '1'
    return env


@pytest.mark.timeout(300)
def test_gpus2_from_a_bare_shell_spawns_its_own_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'],
                         env=_bare_env(), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout          # rank 0 only
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['steps'] == 2 and res['dry_run'] is True


@pytest.mark.timeout(300)
def test_rank_failure_is_the_launchers_exit_code():
    # an invalid flag combination makes every rank exit non-zero; the parent must not report success
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', 'x'],
                         env=_bare_env(), capture_output=True, text=True, timeout=280)
    assert out.returncode != 0


def test_world_size_from_the_environment_wins_over_the_flag(monkeypatch, capsys):
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setenv('RANK', '0')
    assert bench.main(['--gpus', '4', '--dry-run', '--steps', '1']) == 0       # no assert, no spawn
    res = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('{')][0])
    assert res['n_gpus'] == 1


class _Ev:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t


@pytest.mark.parametrize('precision,storage', [('fp32', 'fp32'), ('bf16', 'fp32'), ('bf16', 'bf16')])
def test_roofline_block_follows_the_mode(precision, storage):
    vox = 8 * 128 ** 3
    flop = 2.0 * 32 * 32 * 27 * vox                       # the 32 -> 32 layer at 128^3 x 8
    el = 2.0 if storage == 'bf16' else 4.0
    nbytes = el * 64 * vox
    ms = 6.9 if precision == 'fp32' else 1.4              # round-1 launch times of the two kernels on this layer
    r = bench.roofline_block([(0, _Ev(0.0), _Ev(ms), flop, nbytes)] * 3, precision, storage, dt=1.0)
    assert 0 < r['frac'] <= 1.0 and 0 < r['mfma_frac'] <= 1.0 and 0 < r['hbm_frac'] <= 1.0
    assert r['peak'] in (bench.PEAK_TFLOPS[precision], bench.HBM_PEAK_GBS)
    assert ('bf16' in r['kernel']) == (precision == 'bf16')
    if precision == 'fp32':
        assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - flop / 6.9e-3 / 157.3e12) < 1e-3
    if (precision, storage) == ('bf16', 'fp32'):
        assert r['bound'] == 'hbm' and r['unit'] == 'GB/s'    # 256 B per voxel against 55 kFLOP: left of the bf16 ridge


def test_workload_string_names_the_config():
    a = bench.parse_args([])
    assert bench.workload_string(a, True, 64, 1).startswith('configs[1]') and 'fp32' in bench.workload_string(a, True, 64, 1)
    b = bench.parse_args(['--precision', 'bf16', '--storage', 'bf16'])
    w = bench.workload_string(b, True, 64, 1)
    assert w.startswith('configs[2]') and 'batch of 8' in w and b.volumes_per_step == 8


def test_clock_sampler_without_a_device(tmp_path):
    """bench.py's ClockSampler reads hwmon files when the device has them and reports None when it does not (this container)."""
    import importlib.util
    import os
    import time
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with bench.ClockSampler(0) as cs:
        time.sleep(0.06)
    rep = cs.report()
    assert rep is None or rep['sclk_mhz_median'] >= 0
    # with files: medians of what was read
    f = tmp_path / 'freq1_input'; f.write_text('1870000000\n')
    w = tmp_path / 'power1_input'; w.write_text('1250000000\n')
    cs = bench.ClockSampler(0)
    cs.freq, cs.power = str(f), str(w)
    with cs:
        time.sleep(0.12)
    rep = cs.report()
    assert rep['sclk_mhz_median'] == 1870 and rep['socket_power_w_median'] == 1250 and rep['samples'] >= 1
    r = dict(bound='mfma', frac=0.5, peak=bench.PEAK_TFLOPS['f16x2'])
    bench.add_sustained(r, rep)
    assert abs(r['frac_at_sustained_clock'] - 0.5 * 2400 / 1870) < 1e-3 and 'pure_mfma_stream' in r
