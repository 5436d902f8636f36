"""GPU parity of the Stage-II glue (reference deepwmh/pipeline/DCNN_multistage.py:102-125, metrics.py:26-32) against the
numpy restatement: element-wise fp32 results must be bit-identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _softmaxes(shape, E, seed):
    rng = np.random.default_rng(seed)
    base = rng.random(shape).astype(np.float32)
    out = []
    for e in range(E):
        s = np.clip(base + 0.2 * rng.standard_normal(shape).astype(np.float32), 0, 1).astype(np.float32)
        s[rng.random(shape) < 0.02] = 0.5                      # exact ties of the < 0.5 test
        out.append(s)
    return out


@pytest.mark.parametrize('shape', [(9, 10, 11), (1, 1, 7), (40, 48, 40)])
def test_softmax_masking_bit_exact(shape):
    from deepwmh_amd import stage2
    from oracle import stage2 as o2
    rng = np.random.default_rng(1)
    x = rng.random(shape).astype(np.float32)
    m = (rng.random(shape) < 0.7).astype(np.float32)
    got, ref = stage2.softmax_masking(x, m), o2.softmax_masking(x, m)
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    with pytest.raises(ValueError):
        stage2.softmax_masking(x, m[..., :-1])


@pytest.mark.parametrize('E', [1, 3, 5])
def test_ensembling_bit_exact(E):
    from deepwmh_amd import stage2
    from oracle import stage2 as o2
    sm = _softmaxes((24, 30, 26), E, 7 + E)
    for phys in ([1.0, 1.0, 1.0], [0.9, 0.9, 3.0], [0.6, 0.6, 0.6]):
        field, lab = stage2.ensemble_softmaxes(sm, phys)
        rfield, rlab = o2.ensembling(sm, phys)
        assert np.array_equal(field, rfield) and field.dtype == np.float32
        assert np.array_equal(lab, rlab) and lab.dtype == rlab.dtype
    with pytest.raises(RuntimeError):
        stage2.ensemble_softmaxes(sm, (1.0, 1.0, 1.0))


def test_hard_dice_and_split():
    from deepwmh_amd import stage2
    from oracle import stage2 as o2
    rng = np.random.default_rng(5)
    a = rng.random((30, 31, 32)).astype(np.float32); b = np.clip(a + 0.3 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    assert stage2.hard_dice_binary(a, b) == o2.hard_dice_binary(a, b)
    z = np.zeros((4, 4, 4), np.float32)
    assert stage2.hard_dice_binary(z, z) == o2.hard_dice_binary(z, z) == 0.0
    pairs = [('c%02d' % i, float(d)) for i, d in enumerate(rng.random(47))]
    assert stage2.pick_validation_cases(pairs) == o2.pick_validation_cases(pairs)
    assert stage2.pick_validation_cases(pairs[:3]) == o2.pick_validation_cases(pairs[:3])
    assert stage2.pick_validation_cases(pairs, 0.2) == o2.pick_validation_cases(pairs, 0.2)


def test_file_level_workers(tmp_path):
    from deepwmh_amd import nifti, stage2
    from oracle import stage2 as o2
    shape = (20, 24, 22)
    sm = _softmaxes(shape, 5, 3)
    rng = np.random.default_rng(8)
    valid = (rng.random(shape) < 0.8).astype(np.float32)
    vm = str(tmp_path / 'valid_mask.nii.gz'); nifti.write(vm, valid, spacing=(1.0, 1.0, 1.0))
    masked = []
    for e, s in enumerate(sm):
        raw = str(tmp_path / ('raw_%d_0.nii.gz' % e)); out = str(tmp_path / ('masked_%d_0.nii.gz' % e))
        nifti.write(raw, s, spacing=(1.0, 1.0, 1.0))
        stage2.softmax_masking_files((raw, vm, out))
        masked.append(out)
        assert np.array_equal(nifti.read(out)[0], o2.softmax_masking(s, valid))
    out_file, out_seg = str(tmp_path / 'softmax_ensembled.nii.gz'), str(tmp_path / 'label_ensembled.nii.gz')
    stage2.ensembling_files((masked, None, None, out_file, out_seg, shape, [1.0, 1.0, 1.0]))
    rfield, rlab = o2.ensembling([o2.softmax_masking(s, valid) for s in sm], [1.0, 1.0, 1.0])
    assert np.array_equal(nifti.read(out_file)[0], rfield)
    assert np.array_equal(nifti.read(out_seg)[0], rlab.astype(np.float32))
