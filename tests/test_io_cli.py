"""CPU tests of the boundary's data formats and CLI surface: NIfTI-1, nnU-Net checkpoint layout,
pre/post-processing, and the argv the reference emits."""
import os
import pickle
import shlex

import numpy as np
import pytest

from deepwmh_amd import checkpoint as ck
from deepwmh_amd import nifti
from oracle import preprocess
from deepwmh_amd.cli.predict import build_parser
from deepwmh_amd.plan import tiny_plan


def test_nifti_reference_example_header(golden_dir):
    # first 352 bytes of the reference's examples/Brain_label.nii.gz (data fixture): SURVEY.md 8c
    raw = open(os.path.join(golden_dir, 'brain_label_header.bin'), 'rb').read()
    hdr, swapped = nifti.read_header(raw)
    assert not swapped and hdr['sizeof_hdr'] == 348 and int(hdr['datatype']) == 16 and int(hdr['bitpix']) == 32
    assert list(hdr['dim'][:4]) == [3, 196, 256, 170] and float(hdr['vox_offset']) == 352.0
    assert np.allclose(hdr['pixdim'][1:4], 1.0)


def test_nifti_roundtrip_and_axis_order(tmp_path):
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((5, 7, 9)).astype(np.float32)          # (z, y, x)
    p = str(tmp_path / 'a.nii.gz')
    nifti.write(p, vol, spacing=(3.0, 1.0, 0.5))
    back, meta = nifti.read(p)
    assert np.array_equal(back, vol) and meta['spacing'] == (3.0, 1.0, 0.5) and meta['shape_xyz'] == (9, 7, 5)
    import gzip
    raw = gzip.open(p, 'rb').read()
    assert len(raw) == 352 + vol.size * 4
    assert np.array_equal(np.frombuffer(raw[352:352 + 36], '<f4'), vol[0, 0, :])      # x is the fastest file axis
    seg = (vol > 0).astype(np.uint8)
    p2 = str(tmp_path / 'seg.nii')
    nifti.write(p2, seg, like=meta, dtype=np.uint8)
    back2, meta2 = nifti.read(p2)
    assert np.array_equal(back2, seg.astype(np.float32)) and int(meta2['header']['datatype']) == 2
    assert meta2['spacing'] == meta['spacing']
    with pytest.raises(nifti.NiftiError):
        nifti.read_header(b'\x00' * 348)
    open(str(tmp_path / 'short.nii'), 'wb').write(raw[:400])
    with pytest.raises(nifti.NiftiError):
        nifti.read(str(tmp_path / 'short.nii'))


def test_checkpoint_layout_roundtrip(tmp_path, tiny_net):
    plan, net, sd = tiny_net
    results = str(tmp_path / 'model')
    folder = ck.trainer_folder(results, 'Task002_FinalModel')
    assert folder.endswith('nnUNet/3d_fullres/Task002_FinalModel/nnUNetTrainerV2__nnUNetPlansv2.1')
    plans = ck.make_plans(plan)
    ck.save_checkpoint(folder, sd, plans, 'all', 'model_best', epoch=7)
    assert os.path.isfile(os.path.join(folder, 'all', 'model_best.model'))
    assert os.path.isfile(os.path.join(folder, 'all', 'model_best.model.pkl'))
    assert os.path.isfile(os.path.join(folder, 'plans.pkl'))
    # DeepWMH_install rewrites init[0], init[2], init[3] of the pkl (reference install_model.py:91-106): still loads
    pk = os.path.join(folder, 'all', 'model_best.model.pkl')
    info = pickle.load(open(pk, 'rb'))
    init = list(info['init']); init[0] = '/elsewhere/plans_3D.pkl'; init[2] = '/elsewhere/out'; init[3] = ''
    info['init'] = tuple(init)
    pickle.dump(info, open(pk, 'wb'))
    plan2, plans2, sd2, meta = ck.load_checkpoint(folder, 'all', 'model_best')
    assert meta['epoch'] == 7 and plan2.patch_size == plan.patch_size and plan2.features() == plan.features()
    assert set(sd2) == set(sd) and all(np.array_equal(sd2[k], sd[k]) for k in sd)
    assert ck.resolve_task(results, 'Task002_FinalModel') == 'Task002_FinalModel'
    assert ck.resolve_task(results, '2') == 'Task002_FinalModel'
    with pytest.raises(FileNotFoundError):
        ck.resolve_task(results, '5')
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(folder, 'all', 'model_ep_0046')


def test_preprocess_crop_normalise_export_roundtrip():
    rng = np.random.default_rng(1)
    vol = np.zeros((20, 24, 28), np.float32)
    vol[3:17, 5:20, 4:25] = rng.standard_normal((14, 15, 21)).astype(np.float32) * 50 + 300
    plans = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    data, props = preprocess.preprocess_case(vol, (1.0, 1.0, 1.0), plans)
    assert data.shape == (1, 14, 15, 21) and props['bbox'] == [[3, 17], [5, 20], [4, 25]]
    assert abs(float(data.mean())) < 1e-5 and abs(float(data.std()) - 1) < 1e-3
    prob = np.stack([np.full(data.shape[1:], 0.2, np.float32), np.full(data.shape[1:], 0.8, np.float32)])
    seg, pfull = preprocess.export_prediction(prob, props)
    assert seg.shape == vol.shape and seg[3:17, 5:20, 4:25].min() == 1 and seg.sum() == 14 * 15 * 21
    assert pfull[0, 0, 0, 0] == 1.0 and pfull[1, 10, 10, 10] == np.float32(0.8)
    # resampling path: 2 mm -> 1 mm doubles the grid and comes back to the cropped shape on export
    plans2 = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=False)
    data2, props2 = preprocess.preprocess_case(vol, (2.0, 2.0, 2.0), plans2)
    assert data2.shape == (1, 28, 30, 42)
    seg2, _ = preprocess.export_prediction(np.stack([np.zeros(data2.shape[1:], np.float32), np.ones(data2.shape[1:], np.float32)]), props2)
    assert seg2.shape == vol.shape and seg2.sum() == 14 * 15 * 21


def test_cli_accepts_the_reference_command_lines():
    # format strings at reference deepwmh/main/predict.py:153-155 and pipeline/DCNN_multistage.py:331-336,531-533
    p = build_parser()
    cmd1 = 'nnUNet_predict -i %s -o %s -tr %s -m %s -p %s -t %s -f %s -chk %s --disable_post_processing --selected_cases %s ' % \
        ('/o/001_Preprocessed_Images', '/o/002_Segmentations/001_raw', 'nnUNetTrainerV2', '3d_fullres', 'nnUNetPlansv2.1',
         'Task002_FinalModel', 'all', 'model_best', ' '.join(['caseA', 'case_B.1']))
    a = p.parse_args(shlex.split(cmd1)[1:])
    assert a.task_name == 'Task002_FinalModel' and a.chk == 'model_best' and a.selected_cases == ['caseA', 'case_B.1']
    assert a.folds == ['all'] and not a.disable_tta and not a.save_softmax and a.disable_post_processing
    cmd2 = 'nnUNet_predict -i /raw/imagesTr -o /out/ep46 -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 ' \
           '-t Task001_LabelDenoising -f all -chk model_ep_0046 --save_softmax --disable_tta ' \
           '--disable_post_processing --selected_cases  c1   c2 '
    a = p.parse_args(shlex.split(cmd2)[1:])
    assert a.chk == 'model_ep_0046' and a.save_softmax and a.disable_tta and a.selected_cases == ['c1', 'c2']
    with pytest.raises(SystemExit) as e:          # integrity check runs `nnUNet_predict -h` and expects exit 0
        p.parse_args(['-h'])
    assert e.value.code == 0


def test_checkpoint_helper_process_writes_what_save_checkpoint_writes(tmp_path):
    """`nnUNet_train` hands raw snapshots of its flat arenas to a helper process (cli/_ckpt_writer.py); the files it writes load back to the same
    tensors, extras and momentum as a direct `save_checkpoint`, in submission order, and a failing job surfaces in drain()."""
    import pytest
    from deepwmh_amd import checkpoint as ck
    from deepwmh_amd.plan import tiny_plan
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    plans = ck.make_plans(plan)
    order = [('a.conv.weight', (4, 1, 3, 3, 3)), ('a.conv.bias', (4,)), ('tu.0.weight', (8, 4, 2, 2, 2))]
    sizes = [int(np.prod(s)) for _, s in order]
    offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])])
    rng = np.random.default_rng(0)
    flat, mom = rng.standard_normal(int(offs[-1])).astype(np.float32), rng.standard_normal(int(offs[-1])).astype(np.float32)
    folder = ck.trainer_folder(str(tmp_path), 'Task001_X')
    w = ck.CheckpointProcess(order, offs[:-1], plans, str(tmp_path))
    w.snapshot(flat)
    w.save(folder, 'all', 'model_best', 3, dict(deepwmh_amd_best=0.25, deepwmh_amd_ma=None, deepwmh_amd_loss_scale=65536.0, deepwmh_amd_precision='f16x2'))
    w.add_momentum(mom)
    w.save(folder, 'all', 'model_latest', 3, dict(deepwmh_amd_best=0.25), with_momentum=True)
    flat2 = flat + 1
    w.snapshot(flat2)                                   # the first snapshot's files are dropped behind the jobs that use them
    w.save(folder, 'all', 'model_ep_0004', 4, {})
    w.drain()
    assert sorted(f for f in os.listdir(w.dir) if f.endswith('.f32')) == ['params_2.f32']
    import torch
    for name, src, epoch in (('model_best', flat, 3), ('model_latest', flat, 3), ('model_ep_0004', flat2, 4)):
        c = torch.load(os.path.join(folder, 'all', name + '.model'), map_location='cpu', weights_only=False)
        assert c['epoch'] == epoch
        for (k, shape), o, n in zip(order, offs[:-1], sizes):
            assert tuple(c['state_dict'][k].shape) == shape and np.array_equal(c['state_dict'][k].numpy().ravel(), src[o:o + n])
        assert os.path.isfile(os.path.join(folder, 'all', name + '.model.pkl'))
    c = torch.load(os.path.join(folder, 'all', 'model_latest.model'), map_location='cpu', weights_only=False)
    assert np.array_equal(np.asarray(c['deepwmh_amd_momentum']), mom) and c['deepwmh_amd_best'] == 0.25
    assert 'deepwmh_amd_momentum' not in torch.load(os.path.join(folder, 'all', 'model_best.model'), map_location='cpu', weights_only=False)
    w.save('/proc/definitely/not/writable', 'all', 'model_best', 5, {})
    with pytest.raises(RuntimeError):
        w.drain()
    w.close()
    assert not os.path.exists(w.dir)


def test_nifti_read_into_caller_buffers_and_multi_member_gzip(tmp_path):
    """Round 6: read(alloc=...) places the volume in a caller's buffer (nnUNet_predict hands out page-locked ones); .gz files are decompressed
    with one zlib call per member -- our writer's 1 MiB members, a single-member file as nibabel writes, and a truncated stream."""
    import gzip
    from deepwmh_amd import nifti
    rng = np.random.default_rng(3)
    vol = (rng.standard_normal((64, 96, 80)) * 40).round().astype(np.float32)          # 1.9 MB: below the member threshold -> one member
    big = (rng.standard_normal((96, 128, 112)) * 40).round().astype(np.float32)         # 5.5 MB: several members
    for name, v in (('one', vol), ('many', big)):
        p = str(tmp_path / (name + '.nii.gz'))
        nifti.write(p, v, spacing=(1.0, 2.0, 3.0))
        raw = open(p, 'rb').read()
        members = raw.count(b'\x1f\x8b\x08\x00')                                        # (a lower bound on gzip headers is enough here)
        assert (members >= 2) == (name == 'many')
        assert gzip.open(p).read()[352:] == v.tobytes()                                 # any gzip reader sees one stream
        a, meta = nifti.read(p)
        buf = np.full(v.shape, np.nan, np.float32)
        b, _ = nifti.read(p, alloc=lambda shape: buf)
        assert b is buf and np.array_equal(a, v) and np.array_equal(buf, v) and meta['spacing'] == (1.0, 2.0, 3.0)
    single = str(tmp_path / 'single.nii.gz')
    with gzip.open(single, 'wb', compresslevel=1) as f:
        f.write(gzip.open(str(tmp_path / 'many.nii.gz')).read())
    c, _ = nifti.read(single, alloc=lambda shape: np.empty(shape, np.float32))
    assert np.array_equal(c, big)
    i16 = str(tmp_path / 'i16.nii.gz')
    nifti.write(i16, (vol / 4).astype(np.int16), dtype=np.int16)
    d, _ = nifti.read(i16, alloc=lambda shape: np.empty(shape, np.float32))            # other datatypes are converted into the buffer
    assert d.dtype == np.float32 and np.array_equal(d, (vol / 4).astype(np.int16).astype(np.float32))
    cut = str(tmp_path / 'cut.nii.gz')
    open(cut, 'wb').write(open(single, 'rb').read()[:100000])
    with pytest.raises(nifti.NiftiError):
        nifti.read(cut)
