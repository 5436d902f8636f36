"""ExperimentPlanner3D_v21 restatement (deepwmh_amd/planner.py): the cases DeepWMH meets -- ~1 mm isotropic FLAIR and thick-slice
FLAIR (/root/reference/deepwmh/analysis/image_ops.py:354-359 treats anisotropy > 3 as normal input).  Expected values are
what upstream nnU-Net v1 plans for such datasets (hand-derived from the published rules; parity unpinned)."""
import numpy as np

from deepwmh_amd import planner
from deepwmh_amd.plan import UNetPlan


def test_isotropic_dataset_gets_the_baseline_plan():
    rng = np.random.default_rng(0)
    spacings = [np.array([1.0, 1.0, 1.0])] * 12
    sizes = [np.array([150, 180, 150]) + rng.integers(-6, 7, 3) for _ in range(12)]
    target, tf, tb, st = planner.plan_3d_fullres(spacings, sizes, 1, 2)
    assert np.allclose(target, 1.0) and tf == [0, 1, 2] and tb == [0, 1, 2]
    assert st['pool_op_kernel_sizes'] == [[2, 2, 2]] * 5
    assert st['conv_kernel_sizes'] == [[3, 3, 3]] * 6
    assert st['batch_size'] == 2 and not st['do_dummy_2D_data_aug']
    # 128^3-class patch (BASELINE.json configs[1]): all edges multiples of 32, feature-map budget respected
    assert all(int(p) % 32 == 0 for p in st['patch_size'])
    assert planner.approx_vram(list(st['patch_size']), st['num_pool_per_axis'], 32, 320, 1, 2, st['pool_op_kernel_sizes']) <= planner.REF_VOXELS
    plan = UNetPlan(1, 32, 2, st['pool_op_kernel_sizes'], st['conv_kernel_sizes'], 2, 320, [int(v) for v in st['patch_size']])
    plan.check_supported()
    assert plan.isotropic


def test_thick_slice_dataset():
    # 0.9 mm in plane, 5 mm slices, slice axis LAST in the image's own order: the planner moves it to the front
    spacings = [np.array([0.9, 0.9, 5.0])] * 10 + [np.array([0.9, 0.9, 6.0])] * 2
    sizes = [np.array([230, 256, 28])] * 12
    target, tf, tb, st = planner.plan_3d_fullres(spacings, sizes, 1, 2)
    assert tf == [2, 0, 1] and tb == [1, 2, 0]
    assert np.allclose(target, [0.9, 0.9, 5.0])            # 10th percentile of the slice spacings
    assert np.allclose(st['current_spacing'], [5.0, 0.9, 0.9])
    # in-plane-only pooling and [1,3,3] kernels until the in-plane spacing is within a factor 2 of 5 mm (0.9 -> 1.8 -> 3.6)
    assert st['pool_op_kernel_sizes'][:2] == [[1, 2, 2], [1, 2, 2]]
    assert st['conv_kernel_sizes'][:2] == [[1, 3, 3], [1, 3, 3]]
    assert st['conv_kernel_sizes'][2] == [3, 3, 3] and st['conv_kernel_sizes'][-1] == [3, 3, 3]
    assert all(p in ([1, 2, 2], [2, 2, 2]) for p in st['pool_op_kernel_sizes'])
    assert st['do_dummy_2D_data_aug']
    npa = st['num_pool_per_axis']
    assert all(int(p) % 2 ** n == 0 for p, n in zip(st['patch_size'], npa))
    assert int(st['patch_size'][0]) <= 28 + 2 ** npa[0]
    # the target-spacing rule's floor: never finer than the in-plane spacing
    t2 = planner.target_spacing_v21([np.array([2.0, 2.0, 7.0])] * 9 + [np.array([2.0, 2.0, 1.0])] * 3, [np.array([200, 200, 20])] * 12)
    assert t2[2] >= 2.0


def test_pool_and_conv_props_match_a_hand_worked_case():
    npa, pools, kernels, patch, div = planner.get_pool_and_conv_props([3.0, 1.0, 1.0], [40, 190, 160])
    # stage 0: only the two fine axes are within a factor 2 of the finest spacing -> [1,2,2], kernel [1,3,3]
    assert pools[0] == [1, 2, 2] and kernels[0] == [1, 3, 3]
    # stage 1: spacings (3,2,2): all within a factor 2 -> [2,2,2], kernel [3,3,3]
    assert pools[1] == [2, 2, 2] and kernels[1] == [3, 3, 3]
    assert len(kernels) == len(pools) + 1
    assert div == [2 ** n for n in npa] and all(p % d == 0 for p, d in zip(patch, div))
    assert patch[1] == 192 and patch[2] == 160


def test_transposes_are_inverse():
    for t in ([1.0, 1.0, 5.0], [1.0, 5.0, 1.0], [5.0, 1.0, 1.0], [1.0, 1.0, 1.0]):
        tf, tb = planner.transpose_for(np.array(t))
        assert [tf[i] for i in tb] == [0, 1, 2]
        assert np.argmax(np.array(t)[tf]) == 0


def test_every_emitted_plan_is_one_the_hip_path_runs():
    """upstream's planner lets an in-plane axis run out of voxels one pooling before the other ([2,2,1]) or the first
    axis pool alone ([2,1,1]); `plan_stage` re-plans those (logged) so that `nnUNet_train` / `nnUNet_predict` never meet a plan
    they refuse after the cohort has been preprocessed."""
    import itertools
    from deepwmh_amd.plan import UNetPlan
    msgs = []
    for shape in itertools.product([40, 96, 120, 170, 256], [96, 150, 224], [64, 118, 190, 256]):
        for sp in [(1.2, 1.0, 1.2), (1, 1, 1), (3.0, 0.9, 0.9), (5.0, 0.5, 0.5), (0.9, 0.9, 4.0), (1.0, 2.5, 1.0)]:
            _, _, _, st = planner.plan_3d_fullres([sp] * 20, [shape] * 20, 1, 2, log=msgs.append)
            UNetPlan(1, 32, 2, st['pool_op_kernel_sizes'], st['conv_kernel_sizes'], 2, 320, [int(v) for v in st['patch_size']]).check_supported()
            assert len(st['pool_op_kernel_sizes']) >= 1
    # the advisor's reproduction: 120x150x118 at (1.2, 1.0, 1.2) -> upstream ends with [2,2,1]
    _, _, _, st = planner.plan_3d_fullres([(1.2, 1.0, 1.2)] * 20, [(120, 150, 118)] * 20, 1, 2, log=msgs.append)
    assert list(st['patch_size']) == [128, 160, 112] and st['pool_op_kernel_sizes'] == [[2, 2, 2]] * 4
    assert any('re-planned' in m for m in msgs)
    # the deviation is recorded in the plans entry itself, not only logged
    assert st['deviates_from_upstream']['upstream_pool_op_kernel_sizes'][-1] == [2, 2, 1]
    _, _, _, st = planner.plan_3d_fullres([(1, 1, 1)] * 20, [(160, 192, 160)] * 20, 1, 2)
    assert 'deviates_from_upstream' not in st               # an upstream-identical plan carries no flag


def test_unplannable_dataset_fails_before_preprocessing():
    import pytest
    with pytest.raises(ValueError):
        planner.plan_3d_fullres([(5.0, 1.0, 2.5)] * 4, [(30, 200, 90)] * 4, 1, 2)      # in-plane spacings 2.5 x apart: [1,3,1] kernels
