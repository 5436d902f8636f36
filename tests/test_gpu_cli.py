"""GPU end-to-end test of the process-level drop-in: the command line DeepWMH formats
(reference deepwmh/main/predict.py:153-156), launched the way its run_shell does
(Popen(shlex.split(cmd), shell=False), deepwmh/utilities/external_call.py:50-52), against the oracle."""
import os
import shlex
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nnunet_predict_cli_end_to_end(tmp_path, tiny_net, tiny_io):
    from deepwmh_amd import checkpoint as ck
    from deepwmh_amd import nifti
    from oracle import preprocess
    from oracle import sliding_window as osw
    plan, net, sd = tiny_net
    model_dir = str(tmp_path / 'model')
    plans = ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    folder = ck.trainer_folder(model_dir, 'Task002_FinalModel')
    ck.save_checkpoint(folder, sd, plans, 'all', 'model_best', epoch=100)
    img_dir, out_dir = str(tmp_path / '001_Preprocessed_Images'), str(tmp_path / '002_Segmentations' / '001_raw')
    os.makedirs(img_dir)
    rng = np.random.default_rng(7)
    cases = {}
    for name in ('caseA', 'case_B.1', 'unselected'):
        vol = np.zeros((46, 52, 44), np.float32)
        vol[3:43, 2:50, 2:42] = tiny_io['volume'][0] * 40 + 200 + rng.standard_normal((40, 48, 40)).astype(np.float32)
        nifti.write(os.path.join(img_dir, name + '_0000.nii.gz'), vol, spacing=(1.0, 1.0, 1.0))
        cases[name] = vol
    cmd = 'nnUNet_predict -i %s -o %s -tr %s -m %s -p %s -t %s -f %s -chk %s --disable_post_processing --selected_cases %s ' % \
        (img_dir, out_dir, 'nnUNetTrainerV2', '3d_fullres', 'nnUNetPlansv2.1', 'Task002_FinalModel', 'all', 'model_best',
         ' '.join(['caseA', 'case_B.1'])) + '--save_softmax --disable_tta'
    env = dict(os.environ, RESULTS_FOLDER=model_dir, PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'],
               CUDA_VISIBLE_DEVICES='0')
    p = subprocess.run(shlex.split(cmd), env=env, capture_output=True, text=True)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0
    assert sorted(os.listdir(out_dir)) == ['caseA.nii.gz', 'caseA_0.nii.gz', 'case_B.1.nii.gz', 'case_B.1_0.nii.gz']
    for name in ('caseA', 'case_B.1'):
        seg, meta = nifti.read(os.path.join(out_dir, name + '.nii.gz'))
        p0, _ = nifti.read(os.path.join(out_dir, name + '_0.nii.gz'))
        assert seg.shape == cases[name].shape and int(meta['header']['datatype']) == 2
        data, props = preprocess.preprocess_case(cases[name], (1.0, 1.0, 1.0), plans)
        seg_o, prob_o = osw.predict_3d_tiled(net, data, plan.patch_size, 0.5, False, True)
        seg_ref, prob_ref = preprocess.export_prediction(prob_o, props)
        assert float(np.abs(p0 - prob_ref[0]).max()) <= 1e-4
        sure = np.abs(prob_ref[1] - prob_ref[0]) > 2e-4
        assert int((seg != seg_ref)[sure].sum()) == 0
        assert osw.hard_dice_binary(seg_ref, seg) > 0.9999
    # a missing checkpoint must fail loudly (non-zero exit aborts the reference pipeline)
    bad = subprocess.run(shlex.split(cmd.replace('model_best', 'model_ep_0046')), env=env, capture_output=True, text=True)
    assert bad.returncode != 0


def test_nnunet_predict_fans_out_over_MI355WMH_GPUS(tmp_path, tiny_net, tiny_io):
    """The multi-GPU extension through the executable: MI355WMH_GPUS=0,0 makes the parent start two children (both on
    this box's one GPU) with --part_id 0/1 --num_parts 2; together they must write exactly what one process writes."""
    from deepwmh_amd import checkpoint as ck
    from deepwmh_amd import nifti
    plan, net, sd = tiny_net
    model_dir = str(tmp_path / 'model')
    plans = ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    ck.save_checkpoint(ck.trainer_folder(model_dir, 'Task002_FinalModel'), sd, plans, 'all', 'model_best', epoch=100)
    img_dir = str(tmp_path / 'in')
    os.makedirs(img_dir)
    rng = np.random.default_rng(11)
    names = ['s%02d' % i for i in range(5)]
    for name in names:
        vol = tiny_io['volume'][0] * 40 + 200 + rng.standard_normal((40, 48, 40)).astype(np.float32)
        nifti.write(os.path.join(img_dir, name + '_0000.nii.gz'), vol.astype(np.float32), spacing=(1.0, 1.0, 1.0))
    base = dict(os.environ, RESULTS_FOLDER=model_dir, PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'])
    outs = {}
    for key, extra in (('one', dict(CUDA_VISIBLE_DEVICES='0')), ('fan', dict(MI355WMH_GPUS='0,0'))):
        out_dir = str(tmp_path / key)
        cmd = 'nnUNet_predict -i %s -o %s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t Task002_FinalModel -f all ' \
              '-chk model_best --disable_post_processing' % (img_dir, out_dir)
        env = dict(base, **extra)
        env.pop('MI355WMH_GPUS', None) if key == 'one' else None
        p = subprocess.run(shlex.split(cmd), env=env, capture_output=True, text=True)
        print(p.stdout[-1500:], p.stderr[-1500:])
        assert p.returncode == 0
        if key == 'fan':
            assert 'over 2 GPU(s)' in p.stdout and p.stdout.count('deepwmh_amd nnUNet_predict: task') == 2
        outs[key] = {n: nifti.read(os.path.join(out_dir, n + '.nii.gz'))[0] for n in names}
    for n in names:
        assert np.array_equal(outs['one'][n], outs['fan'][n])
    # a failing part fails the whole command
    bad = subprocess.run(shlex.split(cmd.replace('model_best', 'model_ep_0046')), env=dict(base, MI355WMH_GPUS='0,0'),
                         capture_output=True, text=True)
    assert bad.returncode != 0


def test_nnunet_predict_precision_switches(tmp_path, tiny_net, tiny_io):
    """WMH_PREDICT_PRECISION through the executable: f16x2 (fp32-grade: the label map of the fp32 run up to voxels on the decision
    boundary, probabilities within 1e-4) and --disable_mixed_precision forcing fp32 whatever the environment says."""
    from deepwmh_amd import checkpoint as ck
    from deepwmh_amd import nifti
    plan, net, sd = tiny_net
    model_dir = str(tmp_path / 'model')
    plans = ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    ck.save_checkpoint(ck.trainer_folder(model_dir, 'Task002_FinalModel'), sd, plans, 'all', 'model_best', epoch=100)
    img_dir = str(tmp_path / 'in')
    os.makedirs(img_dir)
    nifti.write(os.path.join(img_dir, 'c0_0000.nii.gz'), (tiny_io['volume'][0] * 40 + 200).astype(np.float32), spacing=(1.0, 1.0, 1.0))
    base = dict(os.environ, RESULTS_FOLDER=model_dir, CUDA_VISIBLE_DEVICES='0', PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'])
    base.pop('MI355WMH_GPUS', None)
    res = {}
    for key, env_extra, flag in (('fp32', dict(WMH_PREDICT_PRECISION='fp32'), ''), ('x3', {}, ''),        # the default is f16x2
                                 ('forced', dict(WMH_PREDICT_PRECISION='bf16', WMH_PREDICT_STORAGE='bf16_ncdhw'), ' --disable_mixed_precision')):
        out_dir = str(tmp_path / key)
        cmd = ('nnUNet_predict -i %s -o %s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t Task002_FinalModel -f all -chk model_best '
               '--save_softmax --disable_post_processing%s' % (img_dir, out_dir, flag))
        p = subprocess.run(shlex.split(cmd), env=dict(base, **env_extra), capture_output=True, text=True)
        print(p.stdout[-800:], p.stderr[-800:])
        assert p.returncode == 0
        res[key] = (nifti.read(os.path.join(out_dir, 'c0.nii.gz'))[0], nifti.read(os.path.join(out_dir, 'c0_0.nii.gz'))[0])
    assert np.array_equal(res['fp32'][0], res['forced'][0]) and np.array_equal(res['fp32'][1], res['forced'][1])
    d = float(np.abs(res['fp32'][1] - res['x3'][1]).max())
    diff = res['fp32'][0] != res['x3'][0]
    assert d <= 5e-5 and (not diff.any() or float(np.abs(res['fp32'][1] - 0.5)[diff].max()) <= 2e-4)
    bad = subprocess.run(shlex.split(cmd.replace(' --disable_mixed_precision', '')), env=dict(base, WMH_PREDICT_PRECISION='fp8'),
                         capture_output=True, text=True)
    assert bad.returncode != 0                                        # unknown precision: fails loudly
