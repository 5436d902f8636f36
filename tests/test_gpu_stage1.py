"""GPU parity of Stage I (reference deepwmh/analysis/lesion_analysis.py:84-278 + image_ops helpers) against the numpy
restatement.  Tolerances: the reference accumulates some means in float32 (np.ma on float32 data) where the device uses
double, so float fields agree to ~1e-6 relative; masks and histograms are exact away from those rounding boundaries."""
import numpy as np
import pytest

from stage1_cases import make_case

pytestmark = pytest.mark.gpu


def test_z_score_and_otsu():
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    x, refs, l1, l2 = make_case()
    m = l1[0]
    got, ref = stage1.z_score(x, m), o.z_score(x, m)
    assert got.dtype == np.float64 and float(np.abs(got - ref).max()) <= 5e-6 * (1 + float(np.abs(ref).max()))
    got, ref = stage1.z_score(x), o.z_score(x)
    assert float(np.abs(got - ref).max()) <= 5e-6 * (1 + float(np.abs(ref).max()))
    rng = np.random.default_rng(2)
    img = np.concatenate([rng.normal(0, 1, 40000), rng.normal(6, 1.5, 25000)]).reshape(50, 1300)
    assert stage1.threshold_otsu(img) == o.threshold_otsu(img)                     # same counts -> same centre
    assert stage1.threshold_otsu(np.full((3, 4, 5), 2.5)) == 2.5


@pytest.mark.parametrize('with_mask', [True, False])
def test_mean_std_grid(with_mask):
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    x, refs, l1, l2 = make_case()
    xz = o.z_score(x, l1[0])
    mask = l1[0] if with_mask else None
    for patch in ([13, 13, 13], [12, 20, 9]):
        gm, gs = stage1.mean_std_grid(xz, patch, mask=mask)
        rm, rs = o.mean_std_grid(xz, patch, mask=mask)
        assert gm.shape == rm.shape
        assert float(np.abs(gm - rm).max()) <= 2e-6 * (1 + float(np.abs(rm).max()))
        assert float(np.abs(gs - rs).max()) <= 2e-5 * (1 + float(np.abs(rs).max()))


@pytest.mark.parametrize('side', [None, '+', '-'])
def test_nll(side):
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    x, refs, l1, l2 = make_case(R=5)
    xz = o.z_score(x, l1[0]); rz = [o.z_score(r, l1[0]) for r in refs]
    for min_std in (0.03, None):
        ga, gm, gs = stage1.nll(xz, rz, min_std=min_std, side=side, return_all=True)
        ra, rm, rs = o.nll(xz, rz, min_std=min_std, side=side, return_all=True)
        assert float(np.abs(gm - rm).max()) <= 1e-12 and float(np.abs(gs - rs).max()) <= 1e-12
        assert float(np.abs(ga - ra).max()) <= 1e-9 * (1 + float(np.abs(ra).max()))
    # use_mask=True (lesion_analysis.py:87-92): every reference counts only above its own Otsu threshold; NaN statistics where
    # none does (the anomaly there is 0 by nan_to_num)
    ga, gm, gs = stage1.nll(xz, rz, min_std=0.03, side=side, return_all=True, use_mask=True)
    ra, rm, rs = o.nll(xz, rz, min_std=0.03, side=side, return_all=True, use_mask=True)
    assert np.array_equal(np.isnan(gm), np.isnan(rm)) and np.array_equal(np.isnan(gs), np.isnan(rs)) and np.isnan(rm).any()
    assert float(np.nanmax(np.abs(gm - rm))) <= 1e-12 and float(np.nanmax(np.abs(gs - rs))) <= 1e-12
    assert float(np.abs(ga - ra).max()) <= 1e-9 * (1 + float(np.abs(ra).max()))


def test_component_filtering_exact():
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    rng = np.random.default_rng(4)
    x, refs, l1, l2 = make_case()
    masks = [l1[0], (rng.random((20, 22, 24)) < 0.7).astype(np.float32), np.zeros((5, 6, 7), np.float32)]
    blobs = np.zeros((18, 20, 16), np.float32); blobs[2:9, 2:9, 2:9] = 1; blobs[10:17, 10:17, 8:15] = 1      # equal-size components: ties
    masks.append(blobs)
    for m in masks:
        for vs in ([1.0, 1.0, 1.0], [0.9, 0.9, 4.0], [3.5, 1.0, 1.0]):
            got, ref = stage1.component_filtering(m, vs), o.component_filtering(m, vs)
            assert got.dtype == ref.dtype and np.array_equal(got, ref), (m.shape, vs)
    # the reference's `erosion` argument is never read by its body (image_ops.py:256-308): same result either way
    assert np.array_equal(stage1.component_filtering(masks[1], [1.0, 1.0, 1.0], erosion=False), o.component_filtering(masks[1], [1.0, 1.0, 1.0]))


def test_median_3mm_and_label_vote_exact():
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    rng = np.random.default_rng(6)
    a = rng.standard_normal((17, 19, 21)) * (rng.random((17, 19, 21)) < 0.6)
    for vs in ([1.0, 1.0, 1.0], [0.8, 0.8, 0.9], [0.9, 0.9, 4.5], [5.0, 0.7, 1.0]):
        assert np.array_equal(stage1.median_3mm(a, vs), o.median_3mm(a, vs)), vs
    # voxels finer than 0.75 mm: windows of 4^3 .. 6^3 voxels, even extents (scipy centres them at k // 2), per-slice 2-D form too
    for vs in ([0.7, 0.7, 0.7], [0.5, 0.5, 0.5], [0.6, 0.45, 0.9], [0.4, 0.5, 2.5]):
        assert np.array_equal(stage1.median_3mm(a, vs), o.median_3mm(a, vs)), vs
    with pytest.raises(Exception):
        stage1.median_3mm(a, [0.3, 0.3, 0.3])                      # 10x10x10 window: beyond the 512-voxel limit, must fail loudly
    labs = [rng.integers(0, 4, (9, 10, 11)).astype(np.float32) for _ in range(5)]
    assert np.array_equal(stage1.average_contiguous_labels(labs), o.average_contiguous_labels(labs))


@pytest.mark.parametrize('prior,vs', [('+', [4.0, 4.0, 4.0]), (None, [4.0, 4.0, 4.0]), ('-', [3.0, 3.0, 13.0])])
def test_nll_analysis_end_to_end(prior, vs):
    from deepwmh_amd import stage1
    from oracle import stage1 as o
    x, refs, l1, l2 = make_case()
    got = stage1.nll_analysis_arrays(x, refs, l1, l2, vs, prior)
    ref = o.nll_analysis_arrays(x, refs, l1, l2, vs, prior)
    assert np.array_equal(got['rough_brain'], ref['rough_brain'])
    assert abs(got['otsu_threshold'] - ref['otsu_threshold']) <= 1e-5 * (1 + abs(ref['otsu_threshold']))
    zi = o.z_score(x, ref['rough_brain'])
    near = np.abs(zi - ref['otsu_threshold']) < 1e-4                # voxels whose Otsu decision sits on the rounding boundary
    assert int((got['valid_mask'] != ref['valid_mask'])[~near].sum()) == 0
    assert float(np.abs(got['normalized_input'] - ref['normalized_input']).max()) <= 1e-5
    scale = 1 + float(np.abs(ref['anomaly']).max())
    assert float(np.abs(got['x_mean'] - ref['x_mean']).max()) <= 1e-5 and float(np.abs(got['x_std'] - ref['x_std']).max()) <= 1e-5
    if not near.any():
        assert float(np.abs(got['anomaly'] - ref['anomaly']).max()) <= 2e-5 * scale
        assert np.allclose(got['curve_x'], ref['curve_x'], rtol=1e-6, atol=1e-9)
        assert float(np.abs(got['curve_y'] - ref['curve_y']).max()) <= 0.05         # log10 counts: a voxel on a bin edge may move
        assert abs(got['threshold'] - ref['threshold']) <= 2.1 * (ref['curve_x'][1] - ref['curve_x'][0])
        sure = np.abs(ref['anomaly'] - ref['threshold']) > 3 * (ref['curve_x'][1] - ref['curve_x'][0])
        assert int((got['segmentation'] != ref['segmentation'])[sure].sum()) == 0
    if prior == '+':
        assert got['segmentation'][18:22, 20:25, 15:19].mean() > 0.9


def test_lesion_analyzer_files(tmp_path):
    from deepwmh_amd import nifti, stage1
    from oracle import postprocess as opp
    x, refs, l1, l2 = make_case(R=4)
    d = str(tmp_path)
    sp = (3.0, 3.0, 3.0)
    paths = {'x': d + '/x.nii.gz', 'r': [], 'm': [], 'y': []}
    nifti.write(paths['x'], x, spacing=sp)
    for i in range(4):
        for key, arr in (('r', refs[i]), ('m', l1[i]), ('y', l2[i])):
            p = '%s/%s%d.nii.gz' % (d, key, i)
            nifti.write(p, arr, spacing=sp)
            paths[key].append(p)
    an = stage1.LesionAnalyzer(d + '/out', num_workers=2)
    an.add_case('case1', paths['x'], paths['r'], paths['m'], paths['y'])
    an.analyze_and_do_segmentation(intensity_prior='+')
    folder = d + '/out/case1'
    for f in ('normalized_input.nii.gz', 'anomaly_score.nii.gz', 'valid_mask.nii.gz', 'preprocessed_image.nii.gz', 'summary.pkl',
              'segmentation.nii.gz', 'segmentation.txt', 'segmentation_pp.nii.gz'):
        assert (tmp_path / 'out' / 'case1' / f).exists(), f
    seg = nifti.read(folder + '/segmentation.nii.gz')[0]
    pp = nifti.read(folder + '/segmentation_pp.nii.gz')[0]
    assert np.array_equal(pp, opp.remove_3mm_sparks(seg, [3.0, 3.0, 3.0]).astype(np.float32))
    assert seg[18:22, 20:25, 15:19].mean() > 0.9
