"""Pre/post steps either side of the network inside `nnUNet_predict` / `nnUNet_plan_and_preprocess`, on the GPU
[nnU-Net, not in tree: preprocessing.GenericPreprocessor.preprocess_test_case / resample_and_normalize,
cropping.crop_to_nonzero, inference.segmentation_export.save_segmentation_nifti_from_softmax]:
crop to the non-zero box, spacing resampling (order 3 data, label-wise order 1 segmentations, order 1 softmax on
export; the low-resolution axis on its own for thick slices), z-score, un-crop -- all in libdeepwmh_amd.so
(`volume_ops.hip`, `resample.hip`).  There is no host (numpy / scipy) form in the product: the CPU restatement these are
checked against lives in oracle/preprocess.py.  Parity unpinned: restated from upstream nnU-Net v1."""
import numpy as np


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

def _export_axis(props):
    """Low-resolution axis (in the input's axis order) for the export resampling, or None."""
    if 'original_spacing' not in props:
        return None
    tb = props['transpose_backward']
    after = np.asarray(props['spacing_after_resampling'], float)[tb]          # back to the input's axis order
    return separate_z_axis(props['original_spacing'], after)

# ---- device path: crop-to-nonzero + z-score on the GPU (no resampling: volumes already at the plan's spacing) ----------
def needs_resampling(shape_zyx, spacing, plans, stage=None):
    stages = plans['plans_per_stage']
    st = stages[stage if stage is not None else (max(stages.keys()) if isinstance(stages, dict) else len(stages) - 1)]
    tf = list(plans.get('transpose_forward', [0, 1, 2]))
    target = np.asarray(st['current_spacing'], float)
    sp = np.asarray(spacing, float)[tf]
    shp = np.asarray(shape_zyx)[tf]
    return bool(np.any(np.round(sp / target * shp).astype(int) != shp))


def _resample_device(x, new_shape, order, axis=None):
    """x: device float32 [C,D,H,W] -> [C,*new_shape] through wmh_resample (double arithmetic, scipy's tap order).  axis: the
    low-resolution axis resampled on its own (nnU-Net's separate-z): first the slices across it with `order` (the same
    kernel with that axis left at its length: an interpolating spline reproduces the samples along an unchanged axis), then
    nearest neighbour along the axis."""
    import torch
    from . import _lib
    if axis is not None and int(new_shape[axis]) != int(x.shape[1 + axis]):
        mid = list(int(v) for v in new_shape)
        mid[axis] = int(x.shape[1 + axis])
        return _resample_device(_resample_device(x, mid, order), new_shape, 0)
    lib = _lib.load()
    Cn, D, H, W = (int(v) for v in x.shape)
    Do, Ho, Wo = (int(v) for v in new_shape)
    if (D, H, W) == (Do, Ho, Wo):
        return x
    out = torch.empty((Cn, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    nws = lib.wmh_resample_workspace_doubles(D, H, W, order)
    ws = torch.empty(max(1, nws), dtype=torch.float64, device=x.device)
    _lib.check(lib.wmh_resample(_lib.ptr(x), _lib.ptr(out), _lib.ptr(ws), Cn, D, H, W, Do, Ho, Wo, order, _lib.current_stream()),
               'wmh_resample')
    return out


def preprocess_case_device(volume_zyx, spacing, plans, stage=None, device='cuda:0'):
    """preprocess_case on the GPU (libdeepwmh_amd.so): non-zero mask -> fill holes (6-connected background components
    that do not reach the border) -> bounding box -> [transpose, order-3 resampling of the data and order-0 of the mask
    when the case is not at the plan's spacing] -> masked z-score (fused with the crop when nothing is resampled).
    Returns (device tensor [C, ...] fp32, props); the only host round trip is the 6-int box."""
    import torch
    from . import _lib
    lib = _lib.load()
    stages = plans['plans_per_stage']
    st_plan = stages[stage if stage is not None else (max(stages.keys()) if isinstance(stages, dict) else len(stages) - 1)]
    tf = list(plans.get('transpose_forward', [0, 1, 2]))
    data = volume_zyx[None] if volume_zyx.ndim == 3 else volume_zyx
    if isinstance(data, torch.Tensor):          # a (page-locked) host tensor or a device tensor: the copy is stream ordered, no staging pass
        x = data.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()
    else:
        x = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(device)
    Cn, D, H, W = (int(v) for v in x.shape)
    n = D * H * W
    st = _lib.current_stream()
    raw = torch.empty(n, dtype=torch.uint8, device=device)
    mask = torch.empty((D, H, W), dtype=torch.uint8, device=device)
    labels = torch.empty(n, dtype=torch.int32, device=device)
    flags = torch.empty(n, dtype=torch.int32, device=device)
    bbox_d = torch.empty(6, dtype=torch.int32, device=device)
    _lib.check(lib.wmh_nonzero_mask(_lib.ptr(x), _lib.ptr(raw), Cn, D, H, W, st), 'wmh_nonzero_mask')
    _lib.check(lib.wmh_fill_holes(_lib.ptr(raw), _lib.ptr(mask), _lib.ptr(labels), _lib.ptr(flags), D, H, W, st), 'wmh_fill_holes')
    _lib.check(lib.wmh_mask_bbox(_lib.ptr(mask), _lib.ptr(bbox_d), D, H, W, st), 'wmh_mask_bbox')
    bb = [int(v) for v in bbox_d.cpu().tolist()]
    bbox = [[bb[0], bb[1]], [bb[2], bb[3]], [bb[4], bb[5]]]
    crop_shape = tuple(b - a for a, b in bbox)
    umn = plans.get('use_mask_for_norm', {0: False})
    use_mask = bool(umn[0] if isinstance(umn, dict) else umn)
    shape_t = tuple(crop_shape[i] for i in tf)                     # crop in network axis order
    sp = np.asarray(spacing, float)[tf]
    target = np.asarray(st_plan['current_spacing'], float)
    new_shape = tuple(int(v) for v in np.round(sp / target * np.array(shape_t)).astype(int))
    stats = torch.empty(Cn, 2, dtype=torch.float32, device=device)
    ws = torch.empty(max(1, lib.wmh_zscore_crop_workspace_doubles(Cn)), dtype=torch.float64, device=device)
    props = dict(original_shape=(D, H, W), bbox=bbox, shape_after_crop=shape_t, transpose_forward=tf,
                 transpose_backward=list(plans.get('transpose_backward', [0, 1, 2])), bbox_device=bbox_d,
                 original_spacing=tuple(float(v) for v in np.asarray(spacing, float)), spacing_after_resampling=tuple(float(v) for v in target))
    if tf == [0, 1, 2] and new_shape == shape_t:
        out = torch.empty((Cn,) + crop_shape, dtype=torch.float32, device=device)
        _lib.check(lib.wmh_zscore_crop(_lib.ptr(x), _lib.ptr(mask), _lib.ptr(bbox_d), _lib.ptr(out), None, _lib.ptr(stats), _lib.ptr(ws),
                                       Cn, D, H, W, int(use_mask), st), 'wmh_zscore_crop')
        return out, props
    sl = tuple(slice(a, b) for a, b in bbox)
    xc = x[(slice(None),) + sl].permute([0] + [i + 1 for i in tf]).contiguous()
    mc = mask[sl].permute(tf).contiguous()
    if new_shape != shape_t:
        xc = _resample_device(xc, new_shape, 3, separate_z_axis(sp, target))
        m2 = torch.empty(new_shape, dtype=torch.uint8, device=device)
        _lib.check(lib.wmh_resample_nearest_u8(_lib.ptr(mc), _lib.ptr(m2), shape_t[0], shape_t[1], shape_t[2], new_shape[0], new_shape[1],
                                               new_shape[2], st), 'wmh_resample_nearest_u8')
        mc = m2
    full = torch.tensor([0, new_shape[0], 0, new_shape[1], 0, new_shape[2]], dtype=torch.int32, device=device)
    out = torch.empty((Cn,) + new_shape, dtype=torch.float32, device=device)
    _lib.check(lib.wmh_zscore_crop(_lib.ptr(xc), _lib.ptr(mc), _lib.ptr(full), _lib.ptr(out), None, _lib.ptr(stats), _lib.ptr(ws),
                                   Cn, new_shape[0], new_shape[1], new_shape[2], int(use_mask), st), 'wmh_zscore_crop')
    return out, props


def export_prediction_device(seg_d, prob_d, props, want_prob0=False, host_alloc=None):
    """(seg uint8, prob fp32 [K,...]) device tensors in network geometry -> (uint8 numpy label map, P(background) numpy or
    None) in the input's geometry: transpose back, order-1 resampling of the probabilities to the cropped shape when it
    differs (then argmax), un-crop into zeros on the device; one D2H copy of the label map.
    host_alloc: optional callable(shape, torch dtype) -> page-locked host tensor; the results are then copied into such tensors (stream
    ordered copies + one synchronisation of the current stream) instead of through `.cpu()`'s pageable staging, and the numpy arrays returned
    are views of them (the caller recycles the tensors once the arrays are written out)."""
    import torch
    from . import _lib
    lib = _lib.load()
    D, H, W = props['original_shape']
    tf, tb = props['transpose_forward'], props['transpose_backward']
    tgt = tuple(int(props['shape_after_crop'][tf.index(i)]) for i in range(3))       # cropped shape in the input's axis order
    need_prob = want_prob0 or tuple(seg_d.shape) != tuple(props['shape_after_crop'])
    if need_prob:
        assert prob_d is not None, 'export_prediction_device: probabilities needed (resampled case or --save_softmax)'
        prob = prob_d.permute([0] + [i + 1 for i in tb]).contiguous()
        if tuple(prob.shape[1:]) != tgt:
            prob = _resample_device(prob, tgt, 1, _export_axis(props))
            seg_c = prob.argmax(0).to(torch.uint8)
        else:
            seg_c = seg_d.permute(tb).contiguous()
    else:
        prob = None
        seg_c = seg_d.permute(tb).contiguous()
    full = torch.empty((D, H, W), dtype=torch.uint8, device=seg_c.device)
    _lib.check(lib.wmh_uncrop_u8(_lib.ptr(seg_c), _lib.ptr(props['bbox_device']), _lib.ptr(full), D, H, W, 0, _lib.current_stream()),
               'wmh_uncrop_u8')
    p0 = None
    if want_prob0:
        p0 = torch.ones((D, H, W), dtype=torch.float32, device=seg_c.device)   # outside the crop everything is background
        sl = tuple(slice(a, b) for a, b in props['bbox'])
        p0[sl] = prob[0]
    if host_alloc is None:
        return full.cpu().numpy(), (p0.cpu().numpy() if p0 is not None else None)
    seg_h = host_alloc((D, H, W), torch.uint8)
    seg_h.copy_(full, non_blocking=True)
    p0_h = None
    if p0 is not None:
        p0_h = host_alloc((D, H, W), torch.float32)
        p0_h.copy_(p0, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return seg_h.numpy(), (p0_h.numpy() if p0_h is not None else None)


def export_segmentation_device(seg_crop, props):
    """Label map only (no resampling / transpose case)."""
    return export_prediction_device(seg_crop, None, props, False)[0]


# ---- training cases (`nnUNet_plan_and_preprocess`): image + segmentation ------------------------------------------------
def crop_training_case_device(data, seg, device='cuda:0'):
    """cropping.crop_to_nonzero for a training case: data [C,z,y,x] float32 and its label map [z,y,x] -> (cropped data,
    cropped segmentation int8 with -1 outside the filled non-zero mask, cropped mask (bool), bbox).  The mask, hole filling
    and box come from the device kernels `preprocess_case_device` uses; numpy only slices."""
    import torch
    from . import _lib
    lib = _lib.load()
    x = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(device)
    Cn, D, H, W = (int(v) for v in x.shape)
    n = D * H * W
    st = _lib.current_stream()
    raw = torch.empty(n, dtype=torch.uint8, device=device)
    mask = torch.empty((D, H, W), dtype=torch.uint8, device=device)
    labels = torch.empty(n, dtype=torch.int32, device=device)
    flags = torch.empty(n, dtype=torch.int32, device=device)
    bbox_d = torch.empty(6, dtype=torch.int32, device=device)
    _lib.check(lib.wmh_nonzero_mask(_lib.ptr(x), _lib.ptr(raw), Cn, D, H, W, st), 'wmh_nonzero_mask')
    _lib.check(lib.wmh_fill_holes(_lib.ptr(raw), _lib.ptr(mask), _lib.ptr(labels), _lib.ptr(flags), D, H, W, st), 'wmh_fill_holes')
    _lib.check(lib.wmh_mask_bbox(_lib.ptr(mask), _lib.ptr(bbox_d), D, H, W, st), 'wmh_mask_bbox')
    bb = [int(v) for v in bbox_d.cpu().tolist()]
    bbox = [[bb[0], bb[1]], [bb[2], bb[3]], [bb[4], bb[5]]]
    sl = tuple(slice(a, b) for a, b in bbox)
    nz = mask.cpu().numpy()[sl] != 0
    seg_c = np.asarray(seg)[sl].astype(np.int8)
    seg_c[~nz] = -1                                      # nnU-Net marks outside-of-body voxels with -1
    return np.asarray(data, dtype=np.float32)[(slice(None),) + sl], seg_c, nz, bbox


def _resample_labels_device(seg_d, new_shape, axis):
    """batchgenerators `resize_segmentation(order=1)` as `resample_patient` calls it (order_seg=1, order_z_seg=0): every
    label's indicator resampled linearly (in the plane of the slices across `axis` when the low-resolution axis goes on its
    own), the label set where that is >= 0.5 in ascending label order, then nearest neighbour along `axis`.
    seg_d: device float32 [D,H,W] holding integer labels.  Returns device float32 [*new_shape]."""
    import torch
    new_shape = tuple(int(v) for v in new_shape)
    if tuple(seg_d.shape) == new_shape:
        return seg_d
    mid = list(new_shape)
    if axis is not None:
        mid[axis] = int(seg_d.shape[axis])
    out = torch.zeros(tuple(mid), dtype=torch.float32, device=seg_d.device)
    for c in torch.unique(seg_d).cpu().tolist():
        ind = (seg_d == c).to(torch.float32)[None].contiguous()
        out[_resample_device(ind, mid, 1)[0] >= 0.5] = c
    if tuple(mid) != new_shape:
        out = _resample_device(out[None].contiguous(), new_shape, 0)[0]
    return out


def resample_normalize_training_case_device(cropped_t, seg_t, nz_t, spacing_t, target_t, use_mask, device='cuda:0'):
    """GenericPreprocessor.resample_and_normalize for one training case, everything in network axis order: order-3 resampling
    of the data (`wmh_resample`), label-wise linear resampling of the segmentation, z-score inside seg >= 0 (or the whole
    volume) by `wmh_zscore_crop`.  Returns (data [C,...] float32, segmentation [...] int8) as numpy arrays."""
    import torch
    from . import _lib
    lib = _lib.load()
    spacing_t, target_t = np.asarray(spacing_t, float), np.asarray(target_t, float)
    shape_t = tuple(int(v) for v in cropped_t.shape[1:])
    new_shape = tuple(int(v) for v in np.round(spacing_t / target_t * np.array(shape_t)).astype(int))
    x = torch.from_numpy(np.ascontiguousarray(cropped_t, dtype=np.float32)).to(device)
    if new_shape != shape_t:
        axis = separate_z_axis(spacing_t, target_t)
        x = _resample_device(x, new_shape, 3, axis)
        seg_d = _resample_labels_device(torch.from_numpy(np.ascontiguousarray(seg_t, dtype=np.float32)).to(device), new_shape, axis)
        seg_r = seg_d.cpu().numpy().astype(np.int8)
        mask = (seg_d >= 0).to(torch.uint8).contiguous()
    else:
        seg_r = np.asarray(seg_t, dtype=np.int8)
        mask = torch.from_numpy(np.ascontiguousarray(nz_t, dtype=np.uint8)).to(device)
    Cn = int(x.shape[0])
    st = _lib.current_stream()
    full = torch.tensor([0, new_shape[0], 0, new_shape[1], 0, new_shape[2]], dtype=torch.int32, device=device)
    out = torch.empty((Cn,) + new_shape, dtype=torch.float32, device=device)
    stats = torch.empty(Cn, 2, dtype=torch.float32, device=device)
    ws = torch.empty(max(1, lib.wmh_zscore_crop_workspace_doubles(Cn)), dtype=torch.float64, device=device)
    _lib.check(lib.wmh_zscore_crop(_lib.ptr(x.contiguous()), _lib.ptr(mask), _lib.ptr(full), _lib.ptr(out), None, _lib.ptr(stats), _lib.ptr(ws),
                                   Cn, new_shape[0], new_shape[1], new_shape[2], int(bool(use_mask)), st), 'wmh_zscore_crop')
    return out.cpu().numpy(), seg_r
