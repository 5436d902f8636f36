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
    # at N > 1 the same command runs the two collective paths north_star names on the same ranks
    # (here: gloo, CPU stand-ins for the tiles / the backward pass) and reports them in `summary`
    c3, c4 = res['summary']['configs3_patch_sharded'], res['summary']['configs4_ddp']
    assert c3['world'] == 2 and c3['bit_identical'] is True and c3['forced'] is False and c3['volumes_per_s'] > 0
    assert c3['exposed_exchange_ms'] >= 0 and len(c3['pred']) == 2          # the round-5 cost model's figures for 2 GPUs, beside the measurement
    assert c4['world'] == 2 and c4['same_params'] is True and c4['ms_per_iteration'] > 0 and len(c4['pred_ms']) == 2


@pytest.mark.timeout(300)
def test_legs_off_keeps_the_collective_free_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '1', '--legs', 'off'],
                         env=_bare_env(), capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert 'summary' not in res


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
    assert ('bf16' in r['kernel'] or 'cl_conv3' in r['kernel']) == (precision == 'bf16')
    assert 'peak_basis' in r or r['bound'] == 'hbm'      # which ceiling the fraction is priced against
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
    assert abs(r['frac_at_sustained_clock'] - 0.5 * 2400 / 1870) < 1e-3


def test_the_printed_line_ends_with_a_summary_that_fits_the_drivers_tail():
    """the driver keeps the last 2000 characters of stdout; every mode's headline number must be inside them."""
    import json
    mode = lambda v, f, c: dict(value=v, roofline=dict(frac=f), clock=dict(sclk_mhz_median=c))
    res = dict(metric='FLAIR volumes/sec sliding-window inference (160x192x160)', precision='f16x2', value=4.8047795683388745, unit='volumes/s',
               n_gpus=1, steps=20, warmup=5, ms_per_step=208.12609314889414, higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32 (tensors, accumulation; 3x3x3 products from fp16 hi+lo pairs, 21-22 significant bits)', data='synthetic',
               config=dict(workload=bench.workload_string(bench.parse_args([]), True, 64, 1), tta=True, patch=[128, 128, 128], step=0.5, batch=16,
                           volumes_per_step=1, weights='random He-init (seed 0)', timed_region='long text that stays in the detail file ' * 5,
                           parallelism='volumes sharded over ranks, no collective'),
               roofline=dict(bound='mfma', kernel=bench.KERNEL_NAME['f16x2'], achieved=334.37, peak=833.3, peak_basis=bench.PEAK_BASIS['f16x2'],
                             unit='TFLOP/s', frac=0.4012, traffic=12733424941, traffic_measured='prose ' * 30, launches=240, avg_launch_ms=7.3987,
                             algorithmic_gflop_per_launch=2473.901, algorithmic_gbytes_per_launch=10.0217, share_of_step_time=0.4266,
                             in_kernel=dict(sclk_mhz=1450, mfma_issue_share=0.65, phases=dict(mfma=0.77, barrier_wait=0.04, epilogue=0.19), layer='prose ' * 20)),
               clock=dict(sclk_mhz_median=1811, socket_power_w_median=1284, samples=80), end_to_end_tflops=293.6, hbm_whole_path=dict(frac=0.15),
               cpu_baseline=dict(value=0.00385619295185118, unit='volumes/s', cores=128, kind='port', sample='whole volume WITHOUT mirroring timed end to end '
                                 '(32.7 s: 8 forwards 32.4 s + aggregation 0.3 s); value = 1 / (that with the forwards x 8 for mirroring)'),
               parity=dict(max_abs_softmax_err=1.1146068572998047e-05, argmax_mismatches=0, argmax_mismatches_any_margin=2),
               parity_vs_fp32_path=dict(max_abs_softmax_diff=3.8e-6, dice_vs_fp32_mask=1.0),
               also=dict(fp32_exact=dict(mode(2.175, 0.882, 2393), parity_vs_oracle=dict(max_abs_softmax_err=1.4e-5, argmax_mismatches=0, argmax_mismatches_any_margin=1)),
                         configs2_bf16_storage=dict(mode(12.97, 0.43, 1960), parity_vs_fp32_path=dict(max_abs_softmax_diff=0.0203, dice_vs_fp32_mask=0.998596)),
                         train_fp32=dict(ms_per_step=59.006, roofline=dict(frac=0.617), parity=dict(gradient_rel_l2_err=1.1e-6)),
                         train_f16x2=dict(ms_per_step=31.779, roofline=dict(frac=0.2163), parity=dict(gradient_rel_l2_err=1.2e-6))),
               cli=dict(predict_mirroring=dict(cases=12, cases_per_s=4.83219, first_case_s=0.31, cases_per_s_after_first=5.1234, command_wall_s=5.1, files_written=12),
                        predict_disable_tta_save_softmax=dict(cases=24, cases_per_s=27.512, first_case_s=0.2, cases_per_s_after_first=31.04, command_wall_s=3.3, files_written=48),
                        train=dict(batches_per_epoch=40, ms_per_iteration_in_epoch=28.91, epoch_wall_s=1.21, ms_per_iteration_all_in=30.25, command_wall_s=6.0),
                        kernel_only=dict(mirroring_volumes_per_s=4.805, no_mirroring_volumes_per_s=38.2, train_ms_f16x2=31.779)),
               legs=dict(configs3_patch_sharded=dict(world=1, volumes=10, volumes_per_s=5.512, ms_per_volume=181.4, exposed_exchange_ms_per_volume_rank0=0.213, rounds_per_volume=8,
                                                     all_gather_bytes_per_round=16777216, bit_identical_to_serial_rank0=True, forced_one_rank_collective=True, predicted=None,
                                                     scaling='strong', dry_run=None),
                         configs4_ddp=dict(world=1, iterations=8, ms_per_iteration=25.612, patches_per_s=78.09, all_reduce_bytes_per_iteration=124812288,
                                           all_reduce_messages_per_iteration=9, parameters_identical_on_all_ranks=True, forced_one_rank_collective=True, predicted=None,
                                           scaling='weak', loss_first_two=[0.66, 0.65], dry_run=None)))
    line = bench.compact_line(res)
    text = json.dumps(line)
    assert list(line)[-1] == 'summary'
    tail = text[-(2000 - 120):]                            # the driver appends a short stderr line to the tail
    start = tail.index('"summary"')
    sm = json.loads(tail[start + len('"summary": '):-1])
    assert sm['configs1_f16x2'] == 4.805 and sm['configs1_fp32_exact'] == 2.175 and sm['configs2_bf16'] == 12.97
    assert sm['train_ms'] == {'fp32': 59.01, 'f16x2': 31.78} and sm['frac']['configs1_f16x2'] == 0.4012
    assert sm['parity_max_abs']['f16x2_vs_oracle'] == 1.11e-05 and sm['parity_max_abs']['fp32_exact_vs_oracle'] == 1.4e-05
    assert sm['cli'] == {'predict_mirroring': [4.832, 5.123, 4.805], 'predict_disable_tta_save_softmax': [27.51, 31.04, 38.2], 'train_ms': [28.91, 30.25, 31.78]}
    assert sm['configs3_patch_sharded'] == dict(world=1, volumes_per_s=5.512, exposed_exchange_ms=0.213, bit_identical=True, forced=True, pred=None)
    assert sm['configs4_ddp'] == dict(world=1, ms_per_iteration=25.612, patches_per_s=78.09, all_reduce_mbytes=124.8, same_params=True, pred_ms=None)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'cpu_baseline'):
        assert key in line
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(line['cpu_baseline'])
    assert len(text) < 3900 and 'prose' not in text
