"""Rows whose code IS in /root/reference (image_ops.py, lesion_analysis.py, metrics.py) against vectors produced by the
REFERENCE's own function bodies (tests/golden/rows_ref.npz, made in the build container by tests/golden/make_golden_ref.py,
which executes the reference's `def`s against real numpy / scipy).  CPU tests: the oracle equals the reference.  GPU test:
the HIP mirrors (deepwmh_amd.postprocess / stage1 / stage2) equal the reference.  Bit-exact for masks, labels and counts;
float fields within the tolerance written at each comparison."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rows_ref.npz'))
SPARKS = (('pp_sparks_iso1', [1.0, 1.0, 1.0]), ('pp_sparks_thick', [0.8, 0.8, 3.0]), ('pp_sparks_fine', [0.6, 0.6, 0.6]),
          ('pp_sparks_half', [0.5, 0.5, 0.5]))
MEDIANS = (('s1_median_iso1', [1.0, 1.0, 1.0]), ('s1_median_fine', [0.7, 0.7, 0.7]), ('s1_median_thick2', [0.9, 0.9, 5.0]),
           ('s1_median_thick0', [6.0, 0.7, 1.0]))
CFS = (('s1_cf_iso', [1.0, 1.0, 1.0]), ('s1_cf_thick0', [4.0, 1.0, 1.0]), ('s1_cf_thick2', [0.9, 0.9, 5.0]))


def _close(got, ref, rel):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    return float(np.abs(got[ok] - ref[ok]).max()) <= rel * (1.0 + float(np.abs(ref[ok]).max()))


def _rz():
    brain = G['s1_brain'].astype('int')
    return brain, G['s1_zscore_masked']


def test_fixture_names_its_sources():
    lines = [str(s) for s in G['sha256']]
    assert sorted(l.split()[1] for l in lines) == ['analysis/image_ops.py', 'analysis/lesion_analysis.py', 'analysis/metrics.py']
    assert all(len(l.split()[0]) == 64 for l in lines)


def test_oracle_postprocess_and_dice_equal_the_reference():
    from oracle import postprocess as opp, stage2 as o2
    m = G['pp_mask'].astype(np.float32)
    for key, vox in SPARKS:
        assert np.array_equal(opp.remove_3mm_sparks(m, vox), G[key]), key
    assert np.array_equal(opp.remove_sparks(m, min_volume=7), G['pp_sparks_min7'])
    from scipy.ndimage import label
    lab, n = label((m > 0.5).astype('int'))
    assert n == int(G['pp_num_parts']) and np.array_equal(lab, G['pp_labeled'])
    assert np.array_equal(opp.components_partition(m) >= 0, G['pp_labeled'] > 0)
    assert o2.hard_dice_binary(G['dice_a'], G['dice_b']) == float(G['dice_ab'])
    assert o2.hard_dice_binary(G['dice_a'], np.zeros_like(G['dice_a'])) == float(G['dice_a0'])
    assert o2.hard_dice_binary(np.zeros_like(G['dice_a']), np.zeros_like(G['dice_a'])) == float(G['dice_00'])


def test_oracle_stage1_functions_equal_the_reference():
    from oracle import stage1 as o1
    brain, xz = _rz()
    x = G['s1_x']
    assert float(o1.masked_mean(x, brain)) == float(G['s1_masked_mean']) and float(o1.masked_std(x, brain)) == float(G['s1_masked_std'])
    assert np.array_equal(o1.z_score(x, mask=brain), G['s1_zscore_masked']) and np.array_equal(o1.z_score(x), G['s1_zscore_plain'])
    rz = [o1.z_score(r, mask=brain) for r in G['s1_refs']]
    assert np.array_equal(np.stack(rz), G['s1_refs_z'])
    for tag, mask in (('a', brain), ('b', brain), ('c', None)):
        mu, sd = o1.mean_std_grid(xz, list(G['s1_grid_%s_patch' % tag]), mask=mask)
        assert np.array_equal(mu, G['s1_grid_%s_mean' % tag]) and np.array_equal(sd, G['s1_grid_%s_std' % tag]), tag
    assert np.array_equal(o1.group_mean(rz), G['s1_group_mean']) and np.array_equal(o1.group_std(rz), G['s1_group_std'])
    masks = [m.astype(np.float32) for m in G['s1_group_masks']]
    assert np.array_equal(o1.group_mean(rz, masks=masks), G['s1_group_mean_masked'], equal_nan=True)
    assert np.array_equal(o1.group_std(rz, masks=masks), G['s1_group_std_masked'], equal_nan=True)
    for side, tag in ((None, 'none'), ('+', 'pos'), ('-', 'neg')):
        an, mu, sg = o1.nll(xz, rz, min_std=0.03, side=side, return_all=True)
        assert np.array_equal(an, G['s1_nll_%s' % tag]), tag
    assert np.array_equal(mu, G['s1_nll_mu']) and np.array_equal(sg, G['s1_nll_sigma'])
    assert np.array_equal(o1.nll(xz, rz, side='+'), G['s1_nll_nominstd'])
    valid = G['s1_valid'].astype('int')
    for key, vox in CFS:
        assert np.array_equal(o1.component_filtering(valid, vox), G[key]), key
    anom = G['s1_nll_pos'] * G['s1_cf_iso']
    for key, vox in MEDIANS:
        assert np.array_equal(o1.median_3mm(anom, vox), G[key]), key
    assert np.array_equal(o1.average_contiguous_labels([l.astype(np.float32) for l in G['s1_l2']]), G['s1_avg_label'])
    an_refs = [o1.nll(s, rz, min_std=0.03, side='+') * valid for s in rz]
    cx, cy, cr, crs, _ = o1.histogram_analysis(anom, an_refs, mask=valid)
    assert np.array_equal(cx, G['s1_hist_x']) and np.array_equal(cy, G['s1_hist_y']) and np.array_equal(cr, G['s1_hist_r'])
    assert np.array_equal(np.stack(crs), G['s1_hist_rs'])


@pytest.mark.gpu
def test_hip_postprocess_and_dice_equal_the_reference():
    from deepwmh_amd import postprocess as pp, stage2
    m = G['pp_mask'].astype(np.float32)
    for key, vox in SPARKS:
        got = pp.remove_3mm_sparks(m, vox)
        assert got.dtype == np.dtype('int') and np.array_equal(got, G[key]), key
    assert np.array_equal(pp.remove_sparks(m, min_volume=7), G['pp_sparks_min7'])
    with pytest.raises(RuntimeError):
        pp.remove_3mm_sparks(m, (1.0, 1.0, 1.0))                     # the reference insists on a list (image_ops.py:350-351)
    assert stage2.hard_dice_binary(G['dice_a'], G['dice_b']) == float(G['dice_ab'])
    assert stage2.hard_dice_binary(G['dice_a'], np.zeros_like(G['dice_a'])) == float(G['dice_a0'])
    assert stage2.hard_dice_binary(np.zeros_like(G['dice_a']), np.zeros_like(G['dice_a'])) == float(G['dice_00'])


@pytest.mark.gpu
def test_hip_stage1_functions_equal_the_reference():
    from deepwmh_amd import stage1
    brain, xz = _rz()
    x = G['s1_x']
    # the reference's masked mean / std run in float32 (numpy.ma on the float32 volume); the device accumulates in double: 1e-6
    assert _close(stage1.z_score(x, mask=brain), G['s1_zscore_masked'], 2e-6)
    assert _close(stage1.z_score(x), G['s1_zscore_plain'], 2e-6)
    for r, rzr in zip(G['s1_refs'], G['s1_refs_z']):
        assert _close(stage1.z_score(r, mask=brain), rzr, 2e-6)
    for tag, mask in (('a', brain), ('b', brain), ('c', None)):
        mu, sd = stage1.mean_std_grid(xz, list(G['s1_grid_%s_patch' % tag]), mask=mask)
        assert _close(mu, G['s1_grid_%s_mean' % tag], 1e-6) and _close(sd, G['s1_grid_%s_std' % tag], 1e-6), tag
    rz_ref = list(G['s1_refs_z'])                                  # the reference's own inputs to nll
    for side, tag in ((None, 'none'), ('+', 'pos'), ('-', 'neg')):
        an, mu, sg = stage1.nll(xz, rz_ref, min_std=0.03, side=side, return_all=True)
        assert _close(an, G['s1_nll_%s' % tag], 1e-9), tag
    assert _close(mu, G['s1_nll_mu'], 1e-12) and _close(sg, G['s1_nll_sigma'], 1e-9)
    assert _close(stage1.nll(xz, rz_ref, side='+'), G['s1_nll_nominstd'], 1e-9)
    valid = G['s1_valid'].astype('int')
    for key, vox in CFS:
        assert np.array_equal(stage1.component_filtering(valid, vox), G[key]), key
    anom = G['s1_nll_pos'] * G['s1_cf_iso']
    for key, vox in MEDIANS:
        assert np.array_equal(stage1.median_3mm(anom, vox), G[key]), key
    assert np.array_equal(stage1.average_contiguous_labels([l.astype(np.float32) for l in G['s1_l2']]), G['s1_avg_label'])
