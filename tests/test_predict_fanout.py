"""`MI355WMH_GPUS` extension of `nnUNet_predict`: the parent starts one child per GPU with
CUDA_VISIBLE_DEVICES=<id> and nnU-Net's --part_id/--num_parts, waits, and fails if any child fails
(/root/reference/deepwmh/utilities/external_call.py:64-72: a non-zero exit aborts the pipeline).  Tested here with a
stub child; tests/test_gpu_cli.py runs the real children on the GPU box."""
import json
import os
import sys
import textwrap

import pytest

from deepwmh_amd.cli import predict as cli

STUB = textwrap.dedent('''
    import json, os, sys, time
    a = sys.argv[1:]
    part, parts = int(a[a.index('--part_id') + 1]), int(a[a.index('--num_parts') + 1])
    log = os.environ['STUB_LOG']
    with open(os.path.join(log, 'part%d.json' % part), 'w') as f:
        json.dump(dict(argv=a, gpu=os.environ.get('CUDA_VISIBLE_DEVICES'), fan=os.environ.get('MI355WMH_GPUS'), parts=parts,
                       hip=os.environ.get('HIP_VISIBLE_DEVICES'), rocr=os.environ.get('ROCR_VISIBLE_DEVICES')), f)
    if os.environ.get('STUB_FAIL_PART') == str(part):
        sys.exit(7)
    if os.environ.get('STUB_FAIL_PART') is not None:
        time.sleep(30)          # a healthy part that must be stopped when another one fails
    sys.exit(0)
''')


def test_gpu_list_forms():
    assert cli.parse_gpu_list(None) == [] and cli.parse_gpu_list('  ') == []
    assert cli.parse_gpu_list('0-7') == list(range(8))
    assert cli.parse_gpu_list('0,2,5') == [0, 2, 5]
    assert cli.parse_gpu_list('3') == [3]
    assert cli.parse_gpu_list('0-1, 4-5') == [0, 1, 4, 5]
    assert cli.parse_gpu_list('0,0') == [0, 0]                   # two processes on one GPU
    for bad in ('3-2', '-1', 'x'):
        with pytest.raises(ValueError):
            cli.parse_gpu_list(bad)


def test_counting_gpus_for_all(tmp_path, monkeypatch):
    """MI355WMH_GPUS=all: the runtime is asked in a short-lived child (never in the fan-out parent); without one, sysfs narrowed by
    ROCR_VISIBLE_DEVICES (that variable used to be ignored).  CUDA_VISIBLE_DEVICES -- which the reference sets to one id
    (predict.py:150-152) -- does not narrow 'all'."""
    topo = tmp_path / 'nodes'
    for i, simd in enumerate([0, 0, 1024, 1024, 1024, 1024]):          # two CPU nodes, four GPUs
        (topo / str(i)).mkdir(parents=True)
        (topo / str(i) / 'properties').write_text('cpu_cores_count 8\nsimd_count %d\n' % simd)
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', '0')
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    assert cli.count_gpus_without_runtime(str(topo), probe_cmd=[]) == 4
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '1,3')
    assert cli.count_gpus_without_runtime(str(topo), probe_cmd=[]) == 2
    probe = tmp_path / 'probe.py'
    probe.write_text('import os\nassert "CUDA_VISIBLE_DEVICES" not in os.environ\nprint("noise")\nprint(3)\n')
    assert cli.count_gpus_without_runtime(str(topo), probe_cmd=[sys.executable, str(probe)]) == 3
    probe.write_text('import sys\nsys.exit(1)\n')                   # probe fails: sysfs
    assert cli.count_gpus_without_runtime(str(topo), probe_cmd=[sys.executable, str(probe)]) == 2


def _stub(tmp_path, monkeypatch):
    stub = tmp_path / 'stub_child.py'
    stub.write_text(STUB)
    log = tmp_path / 'log'
    log.mkdir()
    monkeypatch.setenv('STUB_LOG', str(log))
    monkeypatch.setenv('MI355WMH_GPUS', '0-7')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '0,1,2,3,4,5,6,7')
    return [sys.executable, str(stub)], log


def test_one_child_per_gpu_with_part_ids(tmp_path, monkeypatch):
    child, log = _stub(tmp_path, monkeypatch)
    argv = ['-i', 'in', '-o', 'out', '-t', 'Task002_FinalModel', '-chk', 'model_best', '--selected_cases', 'a', 'b']
    rc = cli.fan_out(argv, [4, 5, 6, 7], ncases=32, child_cmd=child)
    assert rc == 0
    recs = [json.load(open(log / ('part%d.json' % r))) for r in range(4)]
    assert [r['gpu'] for r in recs] == ['4', '5', '6', '7']            # the reference's own single-id convention
    assert all(r['parts'] == 4 and r['fan'] is None and r['hip'] is None for r in recs)     # children do not fan out again
    assert all(r['rocr'] == '0,1,2,3,4,5,6,7' for r in recs)            # the layer below HIP is inherited: ids count within it
    assert all(r['argv'][:len(argv)] == argv for r in recs)             # the reference's flags reach every child untouched


def test_no_more_children_than_cases(tmp_path, monkeypatch):
    child, log = _stub(tmp_path, monkeypatch)
    assert cli.fan_out(['-i', 'x'], list(range(8)), ncases=3, child_cmd=child) == 0
    assert sorted(os.listdir(log)) == ['part0.json', 'part1.json', 'part2.json']


@pytest.mark.timeout(25)
def test_a_failing_child_fails_the_command_and_stops_the_rest(tmp_path, monkeypatch):
    child, log = _stub(tmp_path, monkeypatch)
    monkeypatch.setenv('STUB_FAIL_PART', '2')
    rc = cli.fan_out(['-i', 'x'], [0, 1, 2, 3], ncases=8, child_cmd=child)
    assert rc == 7                                                       # well before the healthy parts' 30 s


def test_list_cases_selected_with_padding(tmp_path):
    for c in ('c1', 'c2', 'c3'):
        (tmp_path / (c + '_0000.nii.gz')).write_bytes(b'')
    (tmp_path / 'notes.txt').write_text('x')
    assert cli.list_cases(str(tmp_path), None) == ['c1', 'c2', 'c3']
    assert cli.list_cases(str(tmp_path), ['', 'c3', '', 'c1']) == ['c1', 'c3']     # DCNN_multistage.py:327-333 pads with spaces
    with pytest.raises(SystemExit):
        cli.list_cases(str(tmp_path), ['nope'])
