"""Which epoch becomes `model_best` -- nnU-Net's rule, since `model_best` is what DeepWMH releases and predicts with
(/root/reference/deepwmh/pipeline/DCNN_multistage.py:529-533, 608-611; deepwmh/main/predict.py:149).

[nnU-Net, not in tree] nnUNetTrainer.run_online_evaluation / finish_online_evaluation, network_trainer.update_eval_criterion_MA /
manage_patience: every epoch runs 50 validation batches; the hard (argmax) true-positive / false-positive / false-negative counts of
each foreground class are summed over all of them (kernel `wmh_online_eval_counts`, full-resolution output only); the epoch's metric is
the mean over the classes of 2 tp / (2 tp + fp + fn) (classes with an empty denominator drop out); the criterion is its moving average
MA <- 0.9 MA + 0.1 metric (first epoch: MA = metric); `model_best` is rewritten when MA exceeds the best MA so far (strictly).

Two deliberate differences, both on the safe side and stated in INTEGRATION.md: (1) nnU-Net's first epoch only initialises the best MA and
writes no `model_best`; this trainer writes one then too, so that the file DeepWMH asks for exists after any run (a one-epoch run, or one
whose MA never improves); (2) when no foreground class has a defined Dice in an epoch (nnU-Net: `np.mean([])` = nan, which poisons the MA
for good) the epoch falls back to nnU-Net's own no-metric branch, MA <- 0.9 MA - 0.1 validation loss."""
import numpy as np

NUM_VAL_BATCHES_PER_EPOCH = 50
ALPHA = 0.9


def epoch_metric(counts, num_classes):
    """counts: the 3 (K - 1) integers of `wmh_online_eval_counts` after the epoch's validation batches (tp | fp | fn per foreground class).
    Returns the mean global foreground Dice, or None when no class has a defined one."""
    c = np.asarray(counts, dtype=np.float64).reshape(3, num_classes - 1)
    den = 2 * c[0] + c[1] + c[2]
    ok = den > 0
    if not ok.any():
        return None
    return float(np.mean(2 * c[0][ok] / den[ok]))


class ModelBestCriterion:
    def __init__(self, ma=None, best=None, alpha=ALPHA):
        """ma / best: the values a `-c` run finds in model_latest (the moving average continues across restarts)."""
        self.ma = None if ma is None else float(ma)
        self.best = None if best is None else float(best)
        self.alpha = float(alpha)

    def update(self, metric, val_loss):
        """End of an epoch.  Returns True when this epoch's parameters are to be written as model_best."""
        x = -float(val_loss) if metric is None else float(metric)
        self.ma = x if self.ma is None else self.alpha * self.ma + (1.0 - self.alpha) * x
        if self.best is None:                   # nnU-Net initialises here without saving; see the module docstring (1)
            self.best = self.ma
            return True
        if self.ma > self.best:
            self.best = self.ma
            return True
        return False
