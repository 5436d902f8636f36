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


def downsample_targets(seg, num_scales):
    """DownsampleSegForDSTransform2 with order 0: nearest-neighbour pick at strides 1, 2, 4, ... seg [N,1,X,Y,Z]."""
    out = []
    for s in range(num_scales):
        k = 2 ** s
        out.append(seg[:, :, ::k, ::k, ::k].contiguous() if k > 1 else seg)
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
