"""Oracle restatement of the nnU-Net v1 training step used by `nnUNet_train ... nnUNetTrainerV2`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

[nnU-Net, not in tree] upstream: training/loss_functions/dice_loss.py (`SoftDiceLoss`, `get_tp_fp_fn_tn`,
`DC_and_CE_loss`), loss_functions/deep_supervision.py (`MultipleOutputLoss2`),
network_training/nnUNetTrainerV2.py (`run_iteration`, `initialize_optimizer_and_scheduler`, `maybe_update_lr`,
deep-supervision weights), utilities/nd_softmax.py.  Reference call sites:
/root/reference/deepwmh/pipeline/DCNN_multistage.py:292-309,501-520 (`nnUNet_train 3d_fullres nnUNetTrainerV2 ...`).
"""
import numpy as np
import torch
import torch.nn.functional as F


def get_tp_fp_fn(net_output, gt, axes):
    """upstream get_tp_fp_fn_tn(net_output, gt, axes, mask=None, square=False) without tn."""
    with torch.no_grad():
        gt = gt.long()
        y_onehot = torch.zeros(net_output.shape, dtype=net_output.dtype)
        y_onehot.scatter_(1, gt, 1)
    tp = net_output * y_onehot
    fp = net_output * (1 - y_onehot)
    fn = (1 - net_output) * y_onehot
    return tp.sum(axes), fp.sum(axes), fn.sum(axes)


def soft_dice_loss(logits, target, batch_dice=True, do_bg=False, smooth=1e-5):
    """SoftDiceLoss(apply_nonlin=softmax_helper, batch_dice, do_bg, smooth).forward"""
    x = torch.softmax(logits, 1)
    axes = ([0] if batch_dice else []) + list(range(2, x.dim()))
    tp, fp, fn = get_tp_fp_fn(x, target, axes)
    nominator = 2 * tp + smooth
    denominator = 2 * tp + fp + fn + smooth
    dc = nominator / (denominator + 1e-8)
    if not do_bg:
        dc = dc[1:] if batch_dice else dc[:, 1:]
    return -dc.mean()


def dc_and_ce_loss(logits, target):
    """DC_and_CE_loss({'batch_dice': True, 'smooth': 1e-5, 'do_bg': False}, {}).forward; target [N,1,...] float labels."""
    dc = soft_dice_loss(logits, target)
    ce = F.cross_entropy(logits, target[:, 0].long())
    return ce + dc


def deep_supervision_weights(num_pool):
    """nnUNetTrainerV2.initialize: 1/2^i, lowest resolution masked out, normalised."""
    w = np.array([1 / (2 ** i) for i in range(num_pool)])
    mask = np.array([True] + [True if i < num_pool - 1 else False for i in range(1, num_pool)])
    w[~mask] = 0
    return w / w.sum()


def multiple_output_loss(outputs, targets, weights):
    """MultipleOutputLoss2.forward"""
    l = weights[0] * dc_and_ce_loss(outputs[0], targets[0])
    for i in range(1, len(outputs)):
        if weights[i] != 0:
            l = l + weights[i] * dc_and_ce_loss(outputs[i], targets[i])
    return l


def resize_segmentation_order0(seg, new_shape):
    """batchgenerators `resize_segmentation(segmentation, new_shape, order=0)` = skimage `resize(seg.astype(float), new_shape,
    0, mode='edge', clip=True, anti_aliasing=False)`, which for order 0 is scipy's `ndimage.zoom(..., order=0, mode='nearest',
    grid_mode=True)`: output voxel o reads input coordinate (o + 0.5) * in / out - 0.5, rounded to the nearest voxel.  For
    a factor 2 that is voxel 2o + 1 (NOT 2o), for 4 it is 4o + 2."""
    from scipy.ndimage import zoom
    seg = np.asarray(seg)
    factors = [n / o for n, o in zip(new_shape, seg.shape)]
    return zoom(seg.astype(float), factors, order=0, mode='nearest', grid_mode=True).astype(seg.dtype)


def downsample_targets(seg, num_scales, pool_op_kernel_sizes=None):
    """nnU-Net `DownsampleSegForDSTransform2(deep_supervision_scales, 0, 0)` [nnUNetTrainerV2.setup_DA_params: scales =
    1 / cumprod(net_num_pool_op_kernel_sizes)]: per scale, `resize_segmentation(seg[b, c], round(shape * scale), order=0)`.
    seg: torch tensor [N, 1, X, Y, Z].  pool_op_kernel_sizes: per pooling [kz, ky, kx] (default 2, 2, 2)."""
    import torch
    pools = pool_op_kernel_sizes or [[2, 2, 2]] * max(0, num_scales - 1)
    out, cum = [seg], np.ones(3)
    a = seg.numpy()
    for s in range(1, num_scales):
        cum = cum * np.asarray(pools[s - 1], dtype=float)
        new_shape = np.round(np.asarray(a.shape[2:]) * (1.0 / cum)).astype(int)
        t = np.stack([np.stack([resize_segmentation_order0(a[b, c], new_shape) for c in range(a.shape[1])]) for b in range(a.shape[0])])
        out.append(torch.from_numpy(t))
    return out


def poly_lr(epoch, max_epochs, initial_lr=1e-2, exponent=0.9):
    return initial_lr * (1 - epoch / max_epochs) ** exponent


def make_optimizer(net, lr=1e-2):
    return torch.optim.SGD(net.parameters(), lr, weight_decay=3e-5, momentum=0.99, nesterov=True)


def run_iteration(net, optimizer, data, target_list, weights, do_backprop=True):
    """nnUNetTrainerV2.run_iteration in fp32 (no AMP): forward (deep supervision) -> loss -> backward ->
    clip_grad_norm_(12) -> optimizer.step().  Returns the loss value."""
    optimizer.zero_grad()
    net.do_ds = True
    output = net(data)
    l = multiple_output_loss(output, target_list, weights)
    if do_backprop:
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
        optimizer.step()
    return float(l.detach())


# ---- which epoch becomes model_best ----------------------------------------------------------------------------------------------
# [nnU-Net, not in tree] network_training/nnUNetTrainer.py (`run_online_evaluation`, `finish_online_evaluation`),
# nnUNetTrainerV2.py (`run_online_evaluation`: full-resolution output / target only), network_trainer.py
# (`update_eval_criterion_MA`, `manage_patience`, val_eval_criterion_alpha = 0.9, num_val_batches_per_epoch = 50).
# Reference call sites: /root/reference/deepwmh/pipeline/DCNN_multistage.py:529-533, 608-611 and deepwmh/main/predict.py:149 (`-chk model_best`).
def run_online_evaluation(output, target):
    """One validation batch: output [N, K, ...] logits (full resolution), target [N, 1, ...] labels.  Returns (tp, fp, fn) arrays of
    K - 1 hard counts summed over the batch."""
    with torch.no_grad():
        num_classes = output.shape[1]
        output_seg = torch.softmax(output, 1).argmax(1)
        target = target[:, 0]
        axes = tuple(range(1, len(target.shape)))
        tp = torch.zeros((target.shape[0], num_classes - 1))
        fp = torch.zeros((target.shape[0], num_classes - 1))
        fn = torch.zeros((target.shape[0], num_classes - 1))
        for c in range(1, num_classes):
            tp[:, c - 1] = ((output_seg == c).float() * (target == c).float()).sum(axes)
            fp[:, c - 1] = ((output_seg == c).float() * (target != c).float()).sum(axes)
            fn[:, c - 1] = ((output_seg != c).float() * (target == c).float()).sum(axes)
        return tp.sum(0).numpy(), fp.sum(0).numpy(), fn.sum(0).numpy()


def finish_online_evaluation(tps, fps, fns):
    """End of an epoch's validation batches: lists of per-batch count arrays -> the epoch's metric (mean global foreground Dice;
    classes that never occurred and were never predicted drop out; `np.mean([])` = nan when none is left)."""
    tp, fp, fn = np.sum(tps, 0), np.sum(fps, 0), np.sum(fns, 0)
    with np.errstate(invalid='ignore', divide='ignore'):
        per_class = [i for i in [2 * i / (2 * i + j + k) for i, j, k in zip(tp, fp, fn)] if not np.isnan(i)]
        return np.mean(per_class)


def update_eval_criterion_MA(ma, all_val_eval_metrics, all_val_losses, alpha=0.9):
    if ma is None:
        return -all_val_losses[-1] if len(all_val_eval_metrics) == 0 else all_val_eval_metrics[-1]
    if len(all_val_eval_metrics) == 0:
        return alpha * ma - (1 - alpha) * all_val_losses[-1]
    return alpha * ma + (1 - alpha) * all_val_eval_metrics[-1]


def model_best_epochs(val_eval_metrics, val_losses=None):
    """Runs `on_epoch_end`'s criterion over a sequence of epochs and returns the (0-based) epochs at which network_trainer.manage_patience
    writes model_best.model: the first epoch only initialises `best_val_eval_criterion_MA`, afterwards a strictly larger MA saves."""
    ma, best, saved, metrics, losses = None, None, [], [], []
    for e, m in enumerate(val_eval_metrics):
        metrics.append(m)
        losses.append(0.0 if val_losses is None else val_losses[e])
        ma = update_eval_criterion_MA(ma, metrics, losses)
        if best is None:
            best = ma
        if ma > best:
            best = ma
            saved.append(e)
    return saved
