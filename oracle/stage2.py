"""TEST INFRASTRUCTURE ONLY -- numpy restatement of DeepWMH's Stage-II glue.

softmax_masking:   /root/reference/deepwmh/pipeline/DCNN_multistage.py:102-109
ensembling:        /root/reference/deepwmh/pipeline/DCNN_multistage.py:111-125
hard_dice_binary:  /root/reference/deepwmh/analysis/metrics.py:26-32
validation split:  /root/reference/deepwmh/pipeline/DCNN_multistage.py:426-446
hard_dice_binary and remove_3mm_sparks are pinned to the reference's own code (tests/golden/rows_ref.npz, tests/test_rows_ref.py).
The masking / ensembling / split expressions live inside NIfTI file workers and a method of the reference (nibabel is absent: they
cannot be executed here) and stay unpinned; they are restated operation for operation on float32 numpy arrays, which is what
load_nifti_simple returns (data_io.py:288-290).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

from .postprocess import remove_3mm_sparks


def softmax_masking(x, m):
    x = np.asarray(x, dtype=np.float32); m = np.asarray(m, dtype=np.float32)
    return 1 - (m * (1 - x))


def ensembling(softmaxes, phys_res):
    shape = np.asarray(softmaxes[0]).shape
    scalar_field = np.zeros(shape).astype('float32')
    for s in softmaxes:
        scalar_field += np.asarray(s, dtype=np.float32)
    scalar_field = scalar_field / len(softmaxes)
    refined_label = (scalar_field < 0.5).astype('float32')
    refined_label = remove_3mm_sparks(refined_label, phys_res)
    return scalar_field, refined_label


def hard_dice_binary(y_true, y_pred):
    assert y_true.shape == y_pred.shape
    y_true = (y_true > 0.5).astype('float32')
    y_pred = (y_pred > 0.5).astype('float32')
    return 2 * np.sum(y_true * y_pred) / (np.sum(y_true) + np.sum(y_pred) + 0.000001)


def pick_validation_cases(case_dice_pairs, validation_set_perc=0.05):
    all_cases = [c for c, _ in case_dice_pairs]
    case_dice_pairs = sorted(case_dice_pairs, reverse=True, key=lambda x: x[1])
    val_target_num = int(len(all_cases) * validation_set_perc)
    if val_target_num < 1:
        val_target_num = 1
    train_cases, val_cases = [], []
    for i in range(len(case_dice_pairs)):
        case, dice = case_dice_pairs[i]
        if len(val_cases) < val_target_num:
            if i % 2 == 0:
                train_cases.append(case)
            else:
                val_cases.append(case)
        else:
            train_cases.append(case)
    return train_cases, val_cases
