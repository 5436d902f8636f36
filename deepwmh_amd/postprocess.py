"""DeepWMH's post-processing of the raw nnU-Net label map, on the GPU.

Mirrors /root/reference/deepwmh/analysis/image_ops.py:325-367 (`remove_sparks`, `remove_3mm_sparks`): same
names, arguments, return type (int array of 0/1) and error behaviour, so the one-line reference-side change is
`from deepwmh_amd.postprocess import remove_3mm_sparks` in deepwmh/main/predict.py:4 (used at :19-26).
The reference labels with scipy.ndimage.label (6-connectivity) and then loops over the components in Python
(one full-volume comparison per component); here the labelling, the component sizes and the filter are three
HBM-bound passes of libdeepwmh_amd.so (wmh_remove_small_components).  There is no CPU fallback.
"""
import numpy as np

from . import _lib


def _device_mask(mask):
    import torch
    if isinstance(mask, torch.Tensor):
        m = (mask > 0.5).to(torch.uint8).contiguous() if mask.dtype != torch.uint8 else (mask != 0).to(torch.uint8).contiguous()
        return m.cuda() if not m.is_cuda else m
    m = np.ascontiguousarray((np.asarray(mask) > 0.5).astype(np.uint8))
    return torch.from_numpy(m).cuda()


def remove_sparks_device(mask_u8, min_volume=3):
    """mask_u8: uint8 device tensor [D,H,W] (non-zero = foreground).  Returns a uint8 device tensor; stream-ordered."""
    import torch
    assert mask_u8.is_cuda and mask_u8.dtype == torch.uint8 and mask_u8.dim() == 3 and mask_u8.is_contiguous()
    lib = _lib.load()
    D, H, W = (int(v) for v in mask_u8.shape)
    out = torch.empty_like(mask_u8)
    if mask_u8.numel() == 0:
        return out
    labels = torch.empty(mask_u8.numel(), dtype=torch.int32, device=mask_u8.device)
    counts = torch.empty_like(labels)
    st = _lib.current_stream()
    _lib.check(lib.wmh_remove_small_components(_lib.ptr(mask_u8), _lib.ptr(out), _lib.ptr(labels), _lib.ptr(counts), D, H, W,
                                               int(min_volume), st), 'wmh_remove_small_components')
    return out


def remove_sparks(mask, min_volume=3, verbose=False):
    """Any component smaller than min_volume is discarded (image_ops.py:325-345).  mask: array thresholded at 0.5."""
    if np.asarray(mask).ndim != 3:
        raise ValueError('remove_sparks: expected a 3-D mask, got shape %s' % (np.asarray(mask).shape,))
    out = remove_sparks_device(_device_mask(mask), min_volume)
    return out.cpu().numpy().astype('int')


def min_volume_for(voxel_size):
    """The component-size threshold remove_3mm_sparks derives from the voxel size (image_ops.py:352-367)."""
    if not isinstance(voxel_size, list):
        raise RuntimeError('voxel_size should be a list of 3 floats.')
    voxel_volume = voxel_size[0] * voxel_size[1] * voxel_size[2]
    voxel_aniso = np.max(voxel_size) / np.min(voxel_size)
    if voxel_aniso > 3.0:
        return 3                                   # thick slices: 3 voxels instead of 3 mm^3
    min_volume = int(np.around(3.0 / voxel_volume))
    return max(min_volume, 2)                      # always remove single voxels


def remove_3mm_sparks(mask, voxel_size):
    """Remove components with volume smaller than 3 mm^3 (image_ops.py:347-367)."""
    return remove_sparks(mask, min_volume=min_volume_for(voxel_size))
