"""Golden vectors for the rows either side of the network: post-processing, Stage-II glue, Stage-I NLL
analysis, resampling.  Inputs are stored (not re-drawn) and the expected outputs come from the numpy/scipy oracle:
they pin OUR restatement of the reference functions (the reference holds no tests for them and is not executed here).

Run from the repo root:  python tests/golden/make_golden_rows.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import postprocess as opp, preprocess, stage1 as o1, stage2 as o2   # noqa: E402
from stage1_cases import make_case                                    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20)
    out = {}
    # remove_3mm_sparks / fill_holes
    m = (rng.random((18, 20, 22)) < 0.22).astype(np.uint8)
    out['pp_mask'] = m
    out['pp_sparks_iso1'] = opp.remove_3mm_sparks(m, [1.0, 1.0, 1.0]).astype(np.uint8)
    out['pp_sparks_thick'] = opp.remove_3mm_sparks(m, [0.8, 0.8, 3.0]).astype(np.uint8)
    out['pp_sparks_fine'] = opp.remove_3mm_sparks(m, [0.6, 0.6, 0.6]).astype(np.uint8)
    shell = np.zeros((14, 15, 16), np.uint8); shell[2:12, 2:13, 2:14] = 1; shell[5:9, 5:10, 5:11] = 0; shell[7, 7, 10:16] = 0
    out['pp_shell'] = shell
    out['pp_shell_filled'] = opp.fill_holes(shell).astype(np.uint8)
    # Stage II
    sm = [np.clip(rng.random((12, 14, 13)) + 0.2 * rng.standard_normal((12, 14, 13)), 0, 1).astype(np.float32) for _ in range(5)]
    valid = (rng.random((12, 14, 13)) < 0.8).astype(np.float32)
    masked = [o2.softmax_masking(s, valid) for s in sm]
    field, label = o2.ensembling(masked, [1.0, 1.0, 1.0])
    out['s2_softmax'] = np.stack(sm); out['s2_valid'] = valid; out['s2_masked'] = np.stack(masked)
    out['s2_field'] = field; out['s2_label'] = label.astype(np.uint8)
    out['s2_dice'] = np.float64(o2.hard_dice_binary(sm[0], sm[1]))
    # resampling (skimage resize semantics)
    vol = (rng.standard_normal((10, 12, 9)) * 30 + 100).astype(np.float32)
    out['rs_in'] = vol
    for order in (0, 1, 3):
        out['rs_out_o%d' % order] = preprocess._resize(vol.astype(np.float64), (15, 9, 13), order).astype(np.float32)
    # Stage I
    x, refs, l1, l2 = make_case((24, 26, 22), 4, 5)
    res = o1.nll_analysis_arrays(x, refs, l1, l2, [4.0, 4.0, 4.0], '+')
    out['s1_x'] = x; out['s1_refs'] = np.stack(refs); out['s1_l1'] = np.stack(l1).astype(np.uint8); out['s1_l2'] = np.stack(l2).astype(np.uint8)
    out['s1_anomaly'] = res['anomaly'].astype(np.float32); out['s1_valid'] = res['valid_mask'].astype(np.uint8)
    out['s1_threshold'] = np.float64(res['threshold']); out['s1_otsu'] = np.float64(res['otsu_threshold'])
    out['s1_segmentation'] = res['segmentation'].astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'rows_f.npz'), **out)
    print('wrote rows_f.npz, %d arrays, %.0f KB' % (len(out), os.path.getsize(os.path.join(HERE, 'rows_f.npz')) / 1e3))


if __name__ == '__main__':
    main()
