"""Stage-II glue of DeepWMH's self-training loop on the GPU.

Mirrors the module-level workers of /root/reference/deepwmh/pipeline/DCNN_multistage.py:
  _parallel_softmax_masking (:102-109)  -> softmax_masking / softmax_masking_files
  _parallel_ensembling      (:111-125)  -> ensemble_softmaxes / ensembling_files
  hard_dice_binary (deepwmh/analysis/metrics.py:26-32) and the Dice-sorted train/validation split (:414-455).
Same argument order and file outputs, so a maintainer swaps the imports; the arithmetic runs in
libdeepwmh_amd.so (element-wise fp32 in numpy's operation order -> bit-identical results).  No CPU fallback.
"""
import numpy as np

from . import _lib, nifti
from .postprocess import min_volume_for, remove_sparks_device


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def softmax_masking(x, m):
    """y = 1-(m*(1-x)); x is the inverted softmax (-> 1 background), m the valid mask."""
    import torch
    lib = _lib.load()
    xd, md = _dev(x), _dev(m)
    if xd.shape != md.shape:
        raise ValueError('softmax_masking: shapes %s and %s differ' % (tuple(xd.shape), tuple(md.shape)))
    y = torch.empty_like(xd)
    _lib.check(lib.wmh_softmax_mask(_lib.ptr(xd), _lib.ptr(md), _lib.ptr(y), xd.numel(), _lib.current_stream()), 'wmh_softmax_mask')
    return y.cpu().numpy()


def ensemble_softmaxes(softmaxes, phys_res):
    """Mean of the (masked, inverted) softmaxes in list order, lesion where it is < 0.5, components under 3 mm^3
    removed.  Returns (scalar_field float32, refined_label int) like the reference writes them."""
    import torch
    lib = _lib.load()
    min_volume = min_volume_for(phys_res)            # raises like remove_3mm_sparks when phys_res is not a list
    stack = torch.stack([_dev(s) for s in softmaxes])
    shape = tuple(stack.shape[1:])
    if len(shape) != 3:
        raise ValueError('ensemble_softmaxes: expected 3-D volumes')
    n = int(np.prod(shape))
    field = torch.empty(shape, dtype=torch.float32, device='cuda')
    label = torch.empty(shape, dtype=torch.uint8, device='cuda')
    _lib.check(lib.wmh_ensemble_softmax(_lib.ptr(stack), len(softmaxes), _lib.ptr(field), _lib.ptr(label), n, _lib.current_stream()),
               'wmh_ensemble_softmax')
    refined = remove_sparks_device(label, min_volume)
    return field.cpu().numpy(), refined.cpu().numpy().astype('int')


def hard_dice_binary(y_true, y_pred):
    """2*sum(t*p)/(sum(t)+sum(p)+1e-6) on the >0.5 binarisations; the three counts come from the GPU (exact integers)."""
    import torch
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    assert y_true.shape == y_pred.shape, 'shapes %s, %s are not match.' % (str(y_true.shape), str(y_pred.shape))
    lib = _lib.load()
    a, b = _dev(y_true), _dev(y_pred)
    out = torch.zeros(3, dtype=torch.int64, device='cuda')
    _lib.check(lib.wmh_binary_overlap(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _lib.current_stream()), 'wmh_binary_overlap')
    tp, st, sp = (np.float32(v) for v in out.cpu().tolist())      # float32 sums of 0/1 values are exact below 2^24
    return 2 * tp / (st + sp + 0.000001)


def pick_validation_cases(case_dice_pairs, validation_set_perc=0.05):
    """Dice-sorted split (DCNN_multistage.py:426-446): highest-Dice cases alternate train/val until the validation
    set holds max(1, int(perc * N)) cases; everything else trains.  Returns (train_cases, val_cases)."""
    pairs = sorted(case_dice_pairs, reverse=True, key=lambda x: x[1])
    val_target_num = max(1, int(len(pairs) * validation_set_perc))
    train_cases, val_cases = [], []
    for i, (case, _) in enumerate(pairs):
        if len(val_cases) < val_target_num:
            (train_cases if i % 2 == 0 else val_cases).append(case)
        else:
            train_cases.append(case)
    return train_cases, val_cases


# ---- file-level workers, argument tuples as the reference builds them ----------------------------------------------------
def softmax_masking_files(params):
    in_path, valid_mask, out_path = params
    x, meta = nifti.read(in_path)
    m, _ = nifti.read(valid_mask)
    nifti.write(out_path, softmax_masking(x, m), like=meta, dtype=np.float32)


def ensembling_files(params):
    in_softmaxes, prior_label, seg_pp, out_file, out_seg, shape, phys_res = params
    arrays, meta = [], None
    for f in in_softmaxes:
        a, mt = nifti.read(f)
        arrays.append(a)
        meta = meta or mt
    field, refined = ensemble_softmaxes(arrays, phys_res)
    nifti.write(out_file, field, like=meta, dtype=np.float32)
    nifti.write(out_seg, refined.astype(np.float32), like=meta, dtype=np.float32)
