"""CPU tests of the Stage-I oracle pieces that restate third-party code: Otsu against an exhaustive search."""
import numpy as np


def test_oracle_otsu_is_the_max_between_class_variance_centre():
    from oracle import stage1 as o
    rng = np.random.default_rng(0)
    img = np.concatenate([rng.normal(0, 1, 4000), rng.normal(6, 1.5, 2500)]).reshape(50, 130)
    thr = o.threshold_otsu(img)
    counts, edges = np.histogram(img.ravel(), bins=256)
    centers = (edges[:-1] + edges[1:]) / 2
    best, best_v = None, -1.0
    for k in range(255):                                          # threshold between bin k and k+1
        w1, w2 = counts[:k + 1].sum(), counts[k + 1:].sum()
        if w1 == 0 or w2 == 0:
            continue
        m1 = (counts[:k + 1] * centers[:k + 1]).sum() / w1
        m2 = (counts[k + 1:] * centers[k + 1:]).sum() / w2
        v = w1 * w2 * (m1 - m2) ** 2
        if v > best_v * (1 + 1e-12):
            best, best_v = centers[k], v
    assert abs(thr - best) <= 1.5 * (edges[1] - edges[0])          # float32 counts may move the argmax by a bin on near-ties
    assert 2.0 < thr < 4.5
    assert o.threshold_otsu(np.full((4, 4), 3.0)) == 3.0


def test_oracle_stage1_runs_and_finds_the_lesion():
    from oracle import stage1 as o
    from stage1_cases import make_case
    x, refs, l1, l2 = make_case()
    out = o.nll_analysis_arrays(x, refs, l1, l2, [4.0, 4.0, 4.0], '+')
    assert out['segmentation'][18:22, 20:25, 15:19].mean() > 0.9
    assert out['segmentation'].sum() < 0.05 * out['valid_mask'].sum()
