"""Which epoch becomes model_best: deepwmh_amd/criterion.py against the oracle's restatement of nnU-Net's
run_online_evaluation / finish_online_evaluation / update_eval_criterion_MA / manage_patience (oracle/training.py).  model_best is what
DeepWMH releases and predicts with: /root/reference/deepwmh/pipeline/DCNN_multistage.py:529-533, 608-611; deepwmh/main/predict.py:149."""
import numpy as np
import pytest
import torch

from deepwmh_amd.criterion import ALPHA, NUM_VAL_BATCHES_PER_EPOCH, ModelBestCriterion, epoch_metric
from oracle import training as ot


def test_constants_are_nnunets():
    assert ALPHA == 0.9 and NUM_VAL_BATCHES_PER_EPOCH == 50


@pytest.mark.parametrize('K', [2, 3])
def test_epoch_metric_is_the_global_foreground_dice_of_the_summed_counts(K):
    g = torch.Generator().manual_seed(K)
    tps, fps, fns = [], [], []
    for b in range(5):
        out = torch.randn(2, K, 6, 7, 8, generator=g)
        tgt = torch.randint(0, K, (2, 1, 6, 7, 8), generator=g).float()
        tp, fp, fn = ot.run_online_evaluation(out, tgt)
        tps.append(tp); fps.append(fp); fns.append(fn)
    want = ot.finish_online_evaluation(tps, fps, fns)
    counts = np.concatenate([np.sum(tps, 0), np.sum(fps, 0), np.sum(fns, 0)]).astype(np.int64)
    assert abs(epoch_metric(counts, K) - want) < 1e-12
    # a class that never occurs and is never predicted drops out of the mean (nnU-Net filters its nan)
    if K == 3:
        c = counts.reshape(3, 2).copy()
        c[:, 1] = 0
        assert abs(epoch_metric(c.reshape(-1), 3) - 2 * c[0, 0] / (2 * c[0, 0] + c[1, 0] + c[2, 0])) < 1e-12
    assert epoch_metric(np.zeros(3 * (K - 1), dtype=np.int64), K) is None       # nnU-Net: np.mean([]) = nan; here: the no-metric branch


def test_model_best_epochs_follow_the_oracle():
    rng = np.random.default_rng(7)
    for trial in range(20):
        metrics = list(np.clip(np.cumsum(rng.normal(0.02, 0.08, 30)), 0, 1))
        want = ot.model_best_epochs(metrics)
        crit, got, mas = ModelBestCriterion(), [], []
        for e, m in enumerate(metrics):
            if crit.update(m, val_loss=1.0 - m):
                got.append(e)
            mas.append(crit.ma)
        # difference (1) of criterion.py: the first epoch also writes model_best (nnU-Net only initialises its best MA there)
        assert got[0] == 0 and got[1:] == want, (trial, got, want)
        ma = None
        for e in range(len(metrics)):
            ma = ot.update_eval_criterion_MA(ma, metrics[:e + 1], [0.0] * (e + 1))
            assert abs(ma - mas[e]) < 1e-12


def test_criterion_continues_across_a_restart_and_without_a_metric():
    metrics = [0.1, 0.3, 0.2, 0.5, 0.4, 0.6]
    a, got_a = ModelBestCriterion(), []
    for m in metrics:
        got_a.append(a.update(m, 0.0))
    b, got_b = ModelBestCriterion(), []
    for m in metrics[:3]:
        got_b.append(b.update(m, 0.0))
    c = ModelBestCriterion(ma=b.ma, best=b.best)            # what `-c` finds in model_latest
    for m in metrics[3:]:
        got_b.append(c.update(m, 0.0))
    assert got_a == got_b and abs(a.ma - c.ma) < 1e-15 and a.best == c.best
    # no foreground class with a defined Dice: nnU-Net's own branch for "no metric", MA <- 0.9 MA - 0.1 val_loss
    d = ModelBestCriterion()
    d.update(None, 0.8)
    assert d.ma == -0.8
    d.update(None, 0.5)
    assert abs(d.ma - ot.update_eval_criterion_MA(-0.8, [], [0.8, 0.5])) < 1e-15
