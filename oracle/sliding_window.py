"""Oracle restatement of nnU-Net v1 sliding-window inference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

[nnU-Net, not in tree] upstream: nnunet/network_architecture/neural_network.py
(`SegmentationNetwork.predict_3D`, `_internal_predict_3D_3Dconv_tiled`,
`_internal_maybe_mirror_and_pred_3D`, `_compute_steps_for_sliding_window`,
`_get_gaussian`) and nnunet/utilities/nd_softmax.py (`softmax_helper`),
batchgenerators `pad_nd_image`.  Reference call sites that fix the options:
/root/reference/deepwmh/main/predict.py:153-156 (mirroring ON, step 0.5, Gaussian),
/root/reference/deepwmh/pipeline/DCNN_multistage.py:331-336 (`--disable_tta`).
"""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """upstream `_compute_steps_for_sliding_window` (np.round = half-to-even)."""
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be >= patch_size"
    assert 0 < step_size <= 1
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step_value = image_size[dim] - patch_size[dim]
        if num_steps[dim] > 1:
            actual = max_step_value / (num_steps[dim] - 1)
        else:
            actual = 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def get_gaussian(patch_size, sigma_scale=1. / 8):
    """upstream `_get_gaussian`: scipy gaussian_filter of a centre delta (4-sigma truncation,
    mode='constant'), /max, float32, zeros -> smallest non-zero."""
    tmp = np.zeros(patch_size)
    center = [i // 2 for i in patch_size]
    sigmas = [i * sigma_scale for i in patch_size]
    tmp[tuple(center)] = 1
    g = gaussian_filter(tmp, sigmas, 0, mode='constant', cval=0)
    g = g / np.max(g) * 1
    g = g.astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_nd_image(image, new_shape):
    """batchgenerators `pad_nd_image(image, new_shape, 'constant', {'constant_values': 0}, True)`
    for a [C, X, Y, Z] array: centred zero pad up to at least new_shape; returns (padded, slicer)."""
    old = np.array(image.shape[-len(new_shape):])
    new = np.array([max(new_shape[i], old[i]) for i in range(len(new_shape))])
    diff = new - old
    below = diff // 2
    above = diff // 2 + diff % 2
    lead = len(image.shape) - len(new_shape)
    pad_list = [[0, 0]] * lead + [[int(below[i]), int(above[i])] for i in range(len(new_shape))]
    if any(p != [0, 0] for p in pad_list):
        res = np.pad(image, pad_list, 'constant', constant_values=0)
    else:
        res = image
    pad_arr = np.array(pad_list)
    pad_arr[:, 1] = np.array(res.shape) - pad_arr[:, 1]
    slicer = tuple(slice(int(a), int(b)) for a, b in pad_arr)
    return res, slicer


def softmax_helper(x):
    return torch.softmax(x, 1)


MIRROR_FLIPS = [(), (4,), (3,), (4, 3), (2,), (4, 2), (3, 2), (4, 3, 2)]  # upstream m = 0..7


def mirror_and_predict(net, x, do_mirroring, mult):
    """upstream `_internal_maybe_mirror_and_pred_3D` with mirror_axes=(0,1,2).

    x: [1, C, X, Y, Z] torch float32; returns [1, K, X, Y, Z] = mean over flips of
    softmax(net(flip(x))) un-flipped, then `*= mult` (the Gaussian map) if given."""
    net_was_ds = getattr(net, 'do_ds', False)
    net.do_ds = False
    result = torch.zeros([1, net.num_classes] + list(x.shape[2:]), dtype=torch.float32)
    flips = MIRROR_FLIPS if do_mirroring else MIRROR_FLIPS[:1]
    num_results = len(flips)
    with torch.no_grad():
        for ax in flips:
            if len(ax) == 0:
                pred = softmax_helper(net(x))
                result += 1 / num_results * pred
            else:
                pred = softmax_helper(net(torch.flip(x, ax)))
                result += 1 / num_results * torch.flip(pred, ax)
    if mult is not None:
        result[:, :] *= mult
    net.do_ds = net_was_ds
    return result


def predict_3d_tiled(net, volume, patch_size, step_size=0.5, do_mirroring=True, use_gaussian=True,
                     patch_hook=None):
    """upstream `_internal_predict_3D_3Dconv_tiled` (all_in_gpu=False, fp32).

    volume: numpy float32 [C, X, Y, Z] (already preprocessed).  Returns
    (segmentation uint8 [X,Y,Z], class_probabilities float32 [K,X,Y,Z])."""
    assert volume.ndim == 4
    patch_size = tuple(int(p) for p in patch_size)
    data, slicer = pad_nd_image(volume, patch_size)
    data_shape = data.shape
    steps = compute_steps_for_sliding_window(patch_size, data_shape[1:], step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    if use_gaussian and num_tiles > 1:
        gmap = get_gaussian(patch_size, 1. / 8)
        add_for_nb_of_preds = gmap
        gmap_t = torch.from_numpy(gmap)
    else:
        gmap_t = None
        add_for_nb_of_preds = np.ones(patch_size, dtype=np.float32)
    K = net.num_classes
    aggregated_results = np.zeros([K] + list(data_shape[1:]), dtype=np.float32)
    aggregated_nb = np.zeros([K] + list(data_shape[1:]), dtype=np.float32)
    data_t = torch.from_numpy(np.ascontiguousarray(data))
    for x in steps[0]:
        for y in steps[1]:
            for z in steps[2]:
                ub = (x + patch_size[0], y + patch_size[1], z + patch_size[2])
                patch = data_t[None, :, x:ub[0], y:ub[1], z:ub[2]]
                pred = mirror_and_predict(net, patch, do_mirroring, gmap_t)[0].numpy()
                if patch_hook is not None:
                    patch_hook((x, y, z), pred)
                aggregated_results[:, x:ub[0], y:ub[1], z:ub[2]] += pred
                aggregated_nb[:, x:ub[0], y:ub[1], z:ub[2]] += add_for_nb_of_preds
    full = (slice(0, K),) + tuple(slicer[1:])
    aggregated_results = aggregated_results[full]
    aggregated_nb = aggregated_nb[full]
    class_probabilities = aggregated_results / aggregated_nb
    seg = class_probabilities.argmax(0).astype(np.uint8)
    return seg, class_probabilities


def hard_dice_binary(y_true, y_pred):
    """/root/reference/deepwmh/analysis/metrics.py:26-32 (restated; defines 'Dice vs ref')."""
    assert y_true.shape == y_pred.shape
    y_true = (y_true > 0.5).astype('float32')
    y_pred = (y_pred > 0.5).astype('float32')
    return 2 * np.sum(y_true * y_pred) / (np.sum(y_true) + np.sum(y_pred) + 0.000001)


def synthetic_flair(shape=(160, 192, 160), seed=0):
    """SURVEY.md 8d synthetic input: blurred noise inside an ellipsoid + bright blobs, z-scored
    over non-zero voxels.  Returns float32 [1, X, Y, Z]."""
    rng = np.random.default_rng(1000 + seed)
    field = gaussian_filter(rng.standard_normal(shape), 4.0).astype(np.float32)
    field /= field.std()
    zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing='ij')
    c = [(s - 1) / 2 for s in shape]
    ax = [0.4375 * shape[0], 0.4427 * shape[1], 0.4375 * shape[2]]  # 70, 85, 70 @ 160x192x160
    mask = (((zz - c[0]) / ax[0]) ** 2 + ((yy - c[1]) / ax[1]) ** 2 + ((xx - c[2]) / ax[2]) ** 2) <= 1
    vol = field.copy()
    for _ in range(int(rng.integers(3, 11))):
        r = float(rng.uniform(2, 6))
        p = [rng.uniform(0.3 * s, 0.7 * s) for s in shape]
        blob = ((zz - p[0]) ** 2 + (yy - p[1]) ** 2 + (xx - p[2]) ** 2) <= r * r
        vol[blob] += 2.0
    vol = vol * mask
    nz = vol[mask]
    vol[mask] = (nz - nz.mean()) / (nz.std() + 1e-8)
    vol[~mask] = 0
    return vol[None].astype(np.float32)
