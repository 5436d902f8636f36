"""TEST INFRASTRUCTURE ONLY -- numpy / scipy restatement of the pre/post steps either side of the network inside
`nnUNet_predict` and `nnUNet_plan_and_preprocess` [nnU-Net v1, not in tree: cropping.crop_to_nonzero,
preprocessing.GenericPreprocessor.resample_and_normalize / resample_patient / resample_data_or_seg,
batchgenerators resize_segmentation, inference.segmentation_export.save_segmentation_nifti_from_softmax]
(SURVEY.md 8f row 1).  skimage's `resize(order, mode='edge', anti_aliasing=False)` -- absent here -- is scipy's
pixel-area-aligned spline zoom (`grid_mode=True, mode='nearest'`), which is what `_resize` calls.
The product's device forms (deepwmh_amd/preprocess.py: preprocess_case_device, export_prediction_device,
preprocess_training_case_device) are checked against these.  Parity unpinned: nnU-Net and skimage are absent from
/root/reference and from this container.  Only tests/, smoke() and bench.py's cpu_baseline leg may import this."""
import numpy as np
from scipy.ndimage import binary_fill_holes, zoom


def create_nonzero_mask(data):
    mask = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        mask |= data[c] != 0
    return binary_fill_holes(mask)


def crop_to_nonzero(data):
    """data [C,z,y,x] -> (cropped data, nonzero mask cropped (bool), bbox [[z0,z1],[y0,y1],[x0,x1]])."""
    mask = create_nonzero_mask(data)
    idx = np.where(mask)
    if len(idx[0]) == 0:
        bbox = [[0, s] for s in data.shape[1:]]
    else:
        bbox = [[int(i.min()), int(i.max()) + 1] for i in idx]
    sl = tuple(slice(a, b) for a, b in bbox)
    return data[(slice(None),) + sl], mask[sl], bbox


def normalize_noncT(data, nonzero_mask, use_mask_for_norm):
    """nnU-Net non-CT scheme: per-channel z-score (inside the non-zero mask when use_mask_for_norm)."""
    out = data.astype(np.float32, copy=True)
    for c in range(out.shape[0]):
        if use_mask_for_norm:
            v = out[c][nonzero_mask]
            out[c][nonzero_mask] = (v - v.mean()) / (v.std() + 1e-8)
            out[c][~nonzero_mask] = 0
        else:
            out[c] = (out[c] - out[c].mean()) / (out[c].std() + 1e-8)
    return out


def _resize(vol, new_shape, order):
    """skimage.transform.resize(order, mode='edge', anti_aliasing=False) == pixel-area-aligned spline zoom."""
    if tuple(vol.shape) == tuple(new_shape):
        return vol
    factors = [n / o for n, o in zip(new_shape, vol.shape)]
    return zoom(vol, factors, order=order, mode='nearest', grid_mode=True, prefilter=order > 1)

SEPARATE_Z_ANISO_THRESHOLD = 3.0     # nnU-Net RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD [preprocessing.py, not in tree]


def separate_z_axis(spacing_a, spacing_b):
    """nnU-Net `resample_patient` with force_separate_z=None [preprocessing.get_do_separate_z / get_lowres_axis]: the axis that
    is resampled on its own (nearest neighbour along it, spline only inside the slices) or None.  Anisotropy (max/min
    spacing > 3) of `spacing_a` decides first (the case's own spacing), then of `spacing_b` (the plan's); the low-resolution
    axis is the one with the largest spacing, and if that is not a single axis (e.g. (0.24, 1.25, 1.25)) everything is
    resampled jointly.  Thick-slice FLAIR is the case the reference treats as normal
    (/root/reference/deepwmh/analysis/image_ops.py:354-359)."""
    for sp in (spacing_a, spacing_b):
        sp = np.asarray(sp, float)
        if sp.max() / sp.min() > SEPARATE_Z_ANISO_THRESHOLD:
            axis = np.where(sp.max() / sp == 1)[0]
            return int(axis[0]) if len(axis) == 1 else None
    return None

def _resize_separate_z(vol, new_shape, order, axis):
    """`resample_data_or_seg(do_separate_z=True, order_z=0)`: every slice across `axis` resized in its plane (order), then
    nearest neighbour along `axis` at the pixel-centre coordinates (o + 0.5) * in / out - 0.5 (map_coordinates order 0)."""
    from scipy.ndimage import map_coordinates
    new_shape = tuple(int(v) for v in new_shape)
    plane = tuple(n for i, n in enumerate(new_shape) if i != axis)
    slices = [_resize(np.take(vol, k, axis=axis), plane, order) for k in range(vol.shape[axis])]
    out = np.stack(slices, axis)
    if out.shape[axis] == new_shape[axis]:
        return out
    grids = np.mgrid[tuple(slice(0, n) for n in new_shape)].astype(np.float64)
    grids[axis] = (grids[axis] + 0.5) * (out.shape[axis] / new_shape[axis]) - 0.5
    return map_coordinates(out, grids, order=0, mode='nearest')


def resample_volume(vol, new_shape, order, axis=None):
    """One channel to new_shape: jointly (axis None) or with the low-resolution axis on its own."""
    if tuple(vol.shape) == tuple(int(v) for v in new_shape):
        return vol
    return _resize(vol, new_shape, order) if axis is None else _resize_separate_z(vol, new_shape, order, axis)


def _resize_labels(seg, new_shape):
    """batchgenerators `resize_segmentation(order=1)` [not in tree]: every label's indicator resized linearly, the label set
    where that is >= 0.5 (labels in ascending order, so the larger label wins a tie)."""
    if tuple(seg.shape) == tuple(int(v) for v in new_shape):
        return seg
    out = np.zeros(tuple(int(v) for v in new_shape), seg.dtype)
    for c in np.unique(seg):
        out[_resize((seg == c).astype(np.float64), new_shape, 1) >= 0.5] = c
    return out


def resample_segmentation(seg, new_shape, axis=None):
    """`resample_patient`'s segmentation branch (order_seg=1, order_z_seg=0): label-wise linear jointly, or in the plane of every
    slice across `axis` followed by nearest neighbour along it."""
    from scipy.ndimage import map_coordinates
    new_shape = tuple(int(v) for v in new_shape)
    if tuple(seg.shape) == new_shape:
        return seg
    if axis is None:
        return _resize_labels(seg, new_shape)
    plane = tuple(n for i, n in enumerate(new_shape) if i != axis)
    out = np.stack([_resize_labels(np.take(seg, k, axis=axis), plane) for k in range(seg.shape[axis])], axis)
    if out.shape[axis] == new_shape[axis]:
        return out
    grids = np.mgrid[tuple(slice(0, n) for n in new_shape)].astype(np.float64)
    grids[axis] = (grids[axis] + 0.5) * (out.shape[axis] / new_shape[axis]) - 0.5
    return map_coordinates(out, grids, order=0, mode='nearest').astype(seg.dtype)


def resample_to_spacing(data, spacing, target_spacing, order=3):
    spacing, target_spacing = np.asarray(spacing, float), np.asarray(target_spacing, float)
    new_shape = np.round(spacing / target_spacing * np.array(data.shape[1:])).astype(int)
    if np.all(new_shape == np.array(data.shape[1:])):
        return data
    axis = separate_z_axis(spacing, target_spacing)
    return np.stack([resample_volume(data[c].astype(np.float64), new_shape, order, axis).astype(np.float32) for c in range(data.shape[0])])


def preprocess_case(volume_zyx, spacing, plans, stage=None):
    """volume_zyx: float32 [z,y,x] (one modality) or [C,z,y,x].  Returns (data [C,...] for the network, props)."""
    data = volume_zyx[None] if volume_zyx.ndim == 3 else volume_zyx
    data = data.astype(np.float32)
    orig_shape = data.shape[1:]
    cropped, nzmask, bbox = crop_to_nonzero(data)
    stages = plans['plans_per_stage']
    st = stages[stage if stage is not None else (max(stages.keys()) if isinstance(stages, dict) else len(stages) - 1)]
    tf = list(plans.get('transpose_forward', [0, 1, 2]))
    target = np.asarray(st['current_spacing'], float)
    cropped = cropped.transpose([0] + [i + 1 for i in tf])
    nzmask = nzmask.transpose(tf)
    sp = np.asarray(spacing, float)[tf]
    shape_after_crop = cropped.shape[1:]
    res = resample_to_spacing(cropped, sp, target, order=3)
    if res.shape[1:] != nzmask.shape:
        nzmask = _resize(nzmask.astype(np.float64), res.shape[1:], 0) > 0.5
    umn = plans.get('use_mask_for_norm', {0: False})
    use_mask = bool(umn[0] if isinstance(umn, dict) else umn)
    out = normalize_noncT(res, nzmask, use_mask)
    props = dict(original_shape=orig_shape, bbox=bbox, shape_after_crop=tuple(shape_after_crop), transpose_forward=tf,
                 transpose_backward=list(plans.get('transpose_backward', [0, 1, 2])),
                 original_spacing=tuple(float(v) for v in np.asarray(spacing, float)), spacing_after_resampling=tuple(float(v) for v in target))
    return np.ascontiguousarray(out), props


def export_prediction(prob, props):
    """prob [K, ...] in network geometry -> (segmentation uint8, probabilities [K]) in the input's geometry:
    transpose back, resample to the cropped shape (order 1), argmax, un-crop into zeros."""
    tb = props['transpose_backward']
    prob = prob.transpose([0] + [i + 1 for i in tb])
    tgt = props['shape_after_crop']
    tgt = tuple(np.array(tgt)[[props['transpose_forward'].index(i) for i in range(3)]]) if tb != [0, 1, 2] else tuple(tgt)
    if prob.shape[1:] != tgt:
        # save_segmentation_nifti_from_softmax [nnU-Net, not in tree]: order 1, and the low-resolution axis on its own (order 0)
        # when the case's spacing (original axis order) or the resampled spacing is anisotropic
        axis = _export_axis(props)
        prob = np.stack([resample_volume(prob[k].astype(np.float64), tgt, 1, axis).astype(np.float32) for k in range(prob.shape[0])])
    seg_c = prob.argmax(0).astype(np.uint8)
    full = np.zeros(props['original_shape'], dtype=np.uint8)
    bb = props['bbox']
    sl = tuple(slice(a, b) for a, b in bb)
    full[sl] = seg_c
    pfull = np.zeros((prob.shape[0],) + tuple(props['original_shape']), dtype=np.float32)
    pfull[0] = 1.0                                  # outside the crop everything is background
    pfull[(slice(None),) + sl] = prob
    return full, pfull

def _export_axis(props):
    """Low-resolution axis (in the input's axis order) for the export resampling, or None."""
    if 'original_spacing' not in props:
        return None
    tb = props['transpose_backward']
    after = np.asarray(props['spacing_after_resampling'], float)[tb]          # back to the input's axis order
    return separate_z_axis(props['original_spacing'], after)
