"""Rows either side of the network against the committed fixtures (tests/golden/rows_f.npz, made by make_golden_rows.py):
CPU tests pin the oracle to the vectors; the GPU tests run the HIP mirrors on the same inputs."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rows_f.npz'))


def test_oracle_reproduces_postprocess_and_stage2_vectors():
    from oracle import postprocess as opp, stage2 as o2
    assert np.array_equal(opp.remove_3mm_sparks(G['pp_mask'], [1.0, 1.0, 1.0]), G['pp_sparks_iso1'])
    assert np.array_equal(opp.remove_3mm_sparks(G['pp_mask'], [0.8, 0.8, 3.0]), G['pp_sparks_thick'])
    assert np.array_equal(opp.remove_3mm_sparks(G['pp_mask'], [0.6, 0.6, 0.6]), G['pp_sparks_fine'])
    assert np.array_equal(opp.fill_holes(G['pp_shell']), G['pp_shell_filled'].astype(bool))
    masked = [o2.softmax_masking(s, G['s2_valid']) for s in G['s2_softmax']]
    assert np.array_equal(np.stack(masked), G['s2_masked'])
    field, label = o2.ensembling(masked, [1.0, 1.0, 1.0])
    assert np.array_equal(field, G['s2_field']) and np.array_equal(label, G['s2_label'])
    assert o2.hard_dice_binary(G['s2_softmax'][0], G['s2_softmax'][1]) == float(G['s2_dice'])


def test_oracle_reproduces_resample_and_stage1_vectors():
    from oracle import preprocess, stage1 as o1
    for order in (0, 1, 3):
        got = preprocess._resize(G['rs_in'].astype(np.float64), (15, 9, 13), order).astype(np.float32)
        assert np.array_equal(got, G['rs_out_o%d' % order])
    res = o1.nll_analysis_arrays(G['s1_x'], list(G['s1_refs']), [a.astype(np.float32) for a in G['s1_l1']],
                                 [a.astype(np.float32) for a in G['s1_l2']], [4.0, 4.0, 4.0], '+')
    assert np.array_equal(res['valid_mask'], G['s1_valid']) and np.array_equal(res['segmentation'], G['s1_segmentation'])
    assert float(np.abs(res['anomaly'] - G['s1_anomaly']).max()) <= 1e-5 * (1 + float(np.abs(G['s1_anomaly']).max()))
    assert abs(res['threshold'] - float(G['s1_threshold'])) <= 1e-9 and abs(res['otsu_threshold'] - float(G['s1_otsu'])) <= 1e-9


@pytest.mark.gpu
def test_hip_mirrors_reproduce_the_vectors():
    import torch
    from deepwmh_amd import postprocess as pp, preprocess, stage1, stage2
    assert np.array_equal(pp.remove_3mm_sparks(G['pp_mask'], [1.0, 1.0, 1.0]), G['pp_sparks_iso1'])
    assert np.array_equal(pp.remove_3mm_sparks(G['pp_mask'], [0.8, 0.8, 3.0]), G['pp_sparks_thick'])
    assert np.array_equal(pp.remove_3mm_sparks(G['pp_mask'], [0.6, 0.6, 0.6]), G['pp_sparks_fine'])
    masked = [stage2.softmax_masking(s, G['s2_valid']) for s in G['s2_softmax']]
    assert np.array_equal(np.stack(masked), G['s2_masked'])
    field, label = stage2.ensemble_softmaxes(masked, [1.0, 1.0, 1.0])
    assert np.array_equal(field, G['s2_field']) and np.array_equal(label, G['s2_label'])
    assert stage2.hard_dice_binary(G['s2_softmax'][0], G['s2_softmax'][1]) == float(G['s2_dice'])
    x = torch.from_numpy(G['rs_in'][None]).cuda()
    for order in (0, 1, 3):
        got = preprocess._resample_device(x, (15, 9, 13), order).cpu().numpy()[0]
        ref = G['rs_out_o%d' % order]
        assert float(np.abs(got - ref).max()) <= (0.0 if order == 0 else 2e-6 * float(np.abs(ref).max()))
    res = stage1.nll_analysis_arrays(G['s1_x'], list(G['s1_refs']), [a.astype(np.float32) for a in G['s1_l1']],
                                     [a.astype(np.float32) for a in G['s1_l2']], [4.0, 4.0, 4.0], '+')
    assert np.array_equal(res['segmentation'], G['s1_segmentation'])
    assert float(np.abs(res['anomaly'] - G['s1_anomaly']).max()) <= 3e-5 * (1 + float(np.abs(G['s1_anomaly']).max()))
    assert abs(res['threshold'] - float(G['s1_threshold'])) <= 1e-4 * (1 + abs(float(G['s1_threshold'])))
