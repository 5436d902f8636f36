"""Training-side boundary: `nnUNet_plan_and_preprocess` / `nnUNet_train` command lines of the reference, the raw /
preprocessed directory contract and the patch loader (CPU), and the full plan -> train -> continue -> predict
chain through the drop-in executables on the GPU."""
import json
import os
import pickle
import shlex
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_raw_dataset(raw_base, task='Task001_LabelDenoising', n=4, shape=(40, 48, 44), seed=0, spacing=(1.0, 1.0, 1.0)):
    """What reference _prepare_nnunet_training_data writes (DCNN_multistage.py:39-93)."""
    from deepwmh_amd import nifti
    from deepwmh_amd.synthetic import synthetic_flair
    root = os.path.join(raw_base, 'nnUNet_raw_data', task)
    os.makedirs(os.path.join(root, 'imagesTr')); os.makedirs(os.path.join(root, 'labelsTr'))
    training = []
    for i in range(n):
        name = 'case_%02d' % i
        vol = synthetic_flair(shape, seed=seed + i)[0]
        img = vol * 40 + 200 * (vol != 0)
        lab = (vol > 1.6).astype(np.float32)
        nifti.write(os.path.join(root, 'imagesTr', name + '_0000.nii.gz'), img, spacing=spacing)
        nifti.write(os.path.join(root, 'labelsTr', name + '.nii.gz'), lab, spacing=spacing)
        training.append({'image': os.path.join(root, 'imagesTr', name + '.nii.gz'), 'label': os.path.join(root, 'labelsTr', name + '.nii.gz')})
    js = {'name': '', 'description': '', 'reference': '', 'license': '', 'release': 'v1.0', 'tensorImageSize': '3D',
          'modality': {'0': 'Modality_01'}, 'labels': {'0': 'background', '1': 'lesion'}, 'numTraining': n, 'numTest': 0,
          'training': training, 'test': []}
    json.dump(js, open(os.path.join(root, 'dataset.json'), 'w'), indent=4)
    return task


@pytest.mark.gpu
def test_plan_and_preprocess_thick_slices(tmp_path):
    """Thick-slice FLAIR (reference image_ops.py:354-359 accepts anisotropy > 3): the slice axis goes first, the stage is stored in
    that order with [1,3,3] / [1,2,2] first levels, and inference preprocessing lands on the same grid as the training data."""
    from deepwmh_amd import nifti
    from deepwmh_amd.data import DATA_ID, plan_and_preprocess
    from oracle import preprocess
    from deepwmh_amd.plan import UNetPlan
    raw, pre = str(tmp_path / 'raw'), str(tmp_path / 'pre')
    # nifti.write takes [z,y,x] arrays and (z,y,x) spacing: 12 slices of 4 mm, 1 mm in plane; slice axis in the MIDDLE here
    task = make_raw_dataset(raw, n=3, shape=(48, 12, 56), spacing=(1.0, 4.0, 1.0))
    t, plans = plan_and_preprocess(raw, pre, '1', True, log=lambda *a: None)
    assert plans['transpose_forward'] == [1, 0, 2] and plans['transpose_backward'] == [1, 0, 2]
    st = plans['plans_per_stage'][0]
    assert np.allclose(st['current_spacing'], [4.0, 1.0, 1.0])
    assert st['pool_op_kernel_sizes'][:2] == [[1, 2, 2], [1, 2, 2]] and st['conv_kernel_sizes'][:2] == [[1, 3, 3], [1, 3, 3]]
    d = np.load(os.path.join(pre, task, DATA_ID + '_stage0', 'case_00.npz'))['data']
    assert d.shape[1] <= 12 and d.shape[2] > 30 and d.shape[3] > 30          # slices first
    plan = UNetPlan.from_nnunet_plans(plans)
    assert plan.pool_op_kernel_sizes[0] == [1, 2, 2] and not plan.isotropic
    vol, meta = nifti.read(os.path.join(raw, 'nnUNet_raw_data', task, 'imagesTr', 'case_00_0000.nii.gz'))
    x, props = preprocess.preprocess_case(vol, meta['spacing'], plans)
    assert x.shape[1:] == d.shape[1:]
    assert np.allclose(x[0], d[0], atol=1e-4)


def test_cli_accepts_reference_training_commands():
    from deepwmh_amd.cli.plan_and_preprocess import build_parser as pp
    from deepwmh_amd.cli.train import build_parser as tp
    a = pp().parse_args(shlex.split('nnUNet_plan_and_preprocess -t %d --verify_dataset_integrity' % 1)[1:])
    assert a.task_ids == ['1'] and a.verify_dataset_integrity
    cmd = 'nnUNet_train %s %s %d %s -e %d -b %d --noval --save_every_epoch --disable_postprocessing_on_folds' % \
        ('3d_fullres', 'nnUNetTrainerV2', 1, 'all', 50, 150) + ' -c'
    a = tp().parse_args(shlex.split(cmd)[1:])
    assert (a.network, a.network_trainer, a.task, a.fold) == ('3d_fullres', 'nnUNetTrainerV2', '1', 'all')
    assert a.epochs == 50 and a.batches_per_epoch == 150 and a.noval and a.save_every_epoch and a.continue_training
    cmd = 'nnUNet_train %s %s %d %s -e %d -b %d --disable_postprocessing_on_folds --custom_val_cases %s ' % \
        ('3d_fullres', 'nnUNetTrainerV2', 2, 'all', 100, 150, ' v1  v_2 ')
    a = tp().parse_args(shlex.split(cmd)[1:])
    assert a.custom_val_cases == ['v1', 'v_2'] and a.epochs == 100 and not a.continue_training
    with pytest.raises(SystemExit) as e:
        tp().parse_args(['-h'])
    assert e.value.code == 0


@pytest.mark.gpu
def test_plan_and_preprocess_and_loader(tmp_path):
    from deepwmh_amd.data import DATA_ID, PatchLoader, plan_and_preprocess
    raw, pre = str(tmp_path / 'raw'), str(tmp_path / 'pre')
    task = make_raw_dataset(raw)
    t, plans = plan_and_preprocess(raw, pre, '1', True, log=lambda *a: None)
    assert t == task
    out = os.path.join(pre, task)
    for f in ('nnUNetPlansv2.1_plans_3D.pkl', 'nnUNetPlansv2.1_plans_2D.pkl', 'dataset_properties.pkl', 'dataset.json'):
        assert os.path.isfile(os.path.join(out, f)), f             # release_model copies these (DCNN_multistage.py:620-631)
    st = plans['plans_per_stage'][0]
    patch = [int(v) for v in st['patch_size']]
    npool = len(st['pool_op_kernel_sizes'])
    assert npool == 3 and all(p % 8 == 0 and 32 <= p <= 48 for p in patch) and plans['num_classes'] == 1
    d = np.load(os.path.join(out, DATA_ID + '_stage0', 'case_00.npz'))['data']
    assert d.shape[0] == 2 and set(np.unique(d[-1]).tolist()) <= {-1.0, 0.0, 1.0}
    body = d[-1] >= 0
    assert abs(float(d[0][body].mean())) < 1e-3 and abs(float(d[0][body].std()) - 1) < 1e-2
    ld = PatchLoader(os.path.join(out, DATA_ID + '_stage0'), ['case_00', 'case_01'], (32, 32, 32), 2, seed=3)
    big = PatchLoader(os.path.join(out, DATA_ID + '_stage0'), ['case_00'], (48, 56, 48), 2, seed=4)     # patch > case: zero pad
    xb, sb = big.next_batch()
    assert xb.shape == (2, 1, 48, 56, 48) and float(np.abs(xb).sum()) > 0
    fg_hits = 0
    for _ in range(6):
        x, s = ld.next_batch()
        assert x.shape == (2, 1, 32, 32, 32) and s.shape == (2, 32, 32, 32) and s.dtype == np.uint8 and s.max() <= 1
        fg_hits += int(s[1].sum() > 0)
    assert fg_hits == 6                                             # the forced-foreground half of every batch
    # integrity check fails loudly on a shape mismatch
    from deepwmh_amd import nifti
    nifti.write(os.path.join(raw, 'nnUNet_raw_data', task, 'labelsTr', 'case_01.nii.gz'), np.zeros((4, 4, 4), np.float32), spacing=(1, 1, 1))
    with pytest.raises(ValueError):
        plan_and_preprocess(raw, pre, '1', True, log=lambda *a: None)


@pytest.mark.gpu
def test_plan_train_continue_predict_through_the_executables(tmp_path):
    raw, pre, res = str(tmp_path / 'raw'), str(tmp_path / 'pre'), str(tmp_path / 'results')
    task = make_raw_dataset(raw, n=3)
    env = dict(os.environ, nnUNet_raw_data_base=raw, nnUNet_preprocessed=pre, RESULTS_FOLDER=res, CUDA_VISIBLE_DEVICES='0',
               PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'])

    def run(cmd):
        p = subprocess.run(shlex.split(cmd), env=env, capture_output=True, text=True)
        print(p.stdout[-1500:], p.stderr[-1500:])
        return p
    assert run('nnUNet_plan_and_preprocess -t 1 --verify_dataset_integrity').returncode == 0
    train = 'nnUNet_train 3d_fullres nnUNetTrainerV2 1 all -e 2 -b 3 --noval --save_every_epoch --disable_postprocessing_on_folds'
    p = run(train)
    assert p.returncode == 0 and 'epoch 2/2' in p.stdout
    fdir = os.path.join(res, 'nnUNet', '3d_fullres', task, 'nnUNetTrainerV2__nnUNetPlansv2.1', 'all')
    have = set(os.listdir(fdir))
    for f in ('model_latest.model', 'model_latest.model.pkl', 'model_best.model', 'model_best.model.pkl', 'model_ep_0001.model',
              'model_ep_0002.model', 'model_ep_0002.model.pkl'):
        assert f in have, f
    assert os.path.isfile(os.path.join(os.path.dirname(fdir), 'plans.pkl'))
    # model_best by nnU-Net's rule (criterion.py): the epochs the log marks are the ones the rule picks from the logged metrics, and
    # model_best.model holds the LAST of them
    import re
    import torch
    from deepwmh_amd.criterion import ModelBestCriterion
    rows = re.findall(r'epoch (\d+)/2: .* val loss (-?[0-9.]+) val dice (n/a|[0-9.]+) MA (-?[0-9.]+)( -> model_best)?', p.stdout)
    assert [int(r[0]) for r in rows] == [1, 2]
    crit, marked = ModelBestCriterion(), []
    for ep, vloss, dice, ma, mark in rows:
        is_best = crit.update(None if dice == 'n/a' else float(dice), float(vloss))
        assert abs(crit.ma - float(ma)) < 2e-4 and is_best == bool(mark)
        if mark:
            marked.append(int(ep))
    assert marked and marked[0] == 1
    assert torch.load(os.path.join(fdir, 'model_best.model'), map_location='cpu', weights_only=False)['epoch'] == marked[-1]
    assert 'precision f16x2' in p.stdout                             # the default mode; the switches that select the others:
    for extra, envx, want in (('--fp32', {}, 'fp32'), ('', dict(WMH_TRAIN_PRECISION='fp32'), 'fp32')):
        q = subprocess.run(shlex.split(train.replace('-e 2', '-e 1').replace('-b 3', '-b 1') + ' ' + extra), env=dict(env, RESULTS_FOLDER=str(tmp_path / ('res_' + want + extra[2:] + ''.join(envx.values()))), **envx),
                           capture_output=True, text=True)
        assert q.returncode == 0 and ('precision ' + want) in q.stdout, (extra, envx, q.stdout[-500:], q.stderr[-500:])
    q = subprocess.run(shlex.split(train), env=dict(env, WMH_TRAIN_PRECISION='bf16'), capture_output=True, text=True)       # removed in round 6
    assert q.returncode != 0 and 'fp32 or f16x2' in (q.stdout + q.stderr)
    p = run(train.replace('-e 2', '-e 3') + ' -c')                  # the reference appends -c when model_latest exists
    assert p.returncode == 0 and 'continuing from' in p.stdout and 'epoch 3/3' in p.stdout and 'epoch 1/3' not in p.stdout
    out = str(tmp_path / 'softmax')
    pr = run('nnUNet_predict -i %s -o %s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t %s -f all -chk model_ep_0002 '
             '--selected_cases case_00 --save_softmax --disable_post_processing --disable_tta'
             % (os.path.join(raw, 'nnUNet_raw_data', task, 'imagesTr'), out, task))
    assert pr.returncode == 0 and sorted(os.listdir(out)) == ['case_00.nii.gz', 'case_00_0.nii.gz']
    from deepwmh_amd import nifti
    p0, _ = nifti.read(os.path.join(out, 'case_00_0.nii.gz'))
    assert p0.shape == (40, 48, 44) and 0.0 <= float(p0.min()) and float(p0.max()) <= 1.0 + 1e-6


@pytest.mark.gpu
def test_thick_slice_dataset_through_the_executables(tmp_path):
    """Thick-slice FLAIR (12 slices of 4 mm, slice axis in the middle of the file's order): planned with [1,3,3] / [1,2,2] first
    stages, trained and predicted; the output comes back on the image's own grid."""
    raw, pre, res = str(tmp_path / 'raw'), str(tmp_path / 'pre'), str(tmp_path / 'results')
    task = make_raw_dataset(raw, n=3, shape=(48, 12, 56), spacing=(1.0, 4.0, 1.0))
    env = dict(os.environ, nnUNet_raw_data_base=raw, nnUNet_preprocessed=pre, RESULTS_FOLDER=res, CUDA_VISIBLE_DEVICES='0',
               PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'])

    def run(cmd):
        p = subprocess.run(shlex.split(cmd), env=env, capture_output=True, text=True)
        print(p.stdout[-1500:], p.stderr[-1500:])
        return p
    assert run('nnUNet_plan_and_preprocess -t 1 --verify_dataset_integrity').returncode == 0
    plans = pickle.load(open(os.path.join(pre, task, 'nnUNetPlansv2.1_plans_3D.pkl'), 'rb'))
    assert plans['transpose_forward'] == [1, 0, 2] and plans['plans_per_stage'][0]['pool_op_kernel_sizes'][0] == [1, 2, 2]
    p = run('nnUNet_train 3d_fullres nnUNetTrainerV2 1 all -e 2 -b 3 --noval --disable_postprocessing_on_folds')
    assert p.returncode == 0 and 'epoch 2/2' in p.stdout
    losses = [float(l.split('train loss')[1].split()[0]) for l in p.stdout.splitlines() if 'train loss' in l]
    assert len(losses) == 2 and all(np.isfinite(losses))
    out = str(tmp_path / 'seg')
    pr = run('nnUNet_predict -i %s -o %s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t %s -f all -chk model_latest '
             '--selected_cases case_00 --disable_post_processing' % (os.path.join(raw, 'nnUNet_raw_data', task, 'imagesTr'), out, task))
    assert pr.returncode == 0
    from deepwmh_amd import nifti
    seg, meta = nifti.read(os.path.join(out, 'case_00.nii.gz'))
    assert seg.shape == (48, 12, 56) and np.allclose(meta['spacing'], (1.0, 4.0, 1.0), atol=1e-4)
    assert set(np.unique(seg).astype(int).tolist()) <= {0, 1}
