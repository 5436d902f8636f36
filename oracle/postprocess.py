"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the whole-volume passes either side of the network.

remove_sparks / remove_3mm_sparks restate /root/reference/deepwmh/analysis/image_ops.py:325-367 with the same
primitive the reference imports (scipy.ndimage.label, default 6-connectivity, image_ops.py:8); the reference's
per-component Python loop is replaced by a bincount of the label image (same set of kept voxels).
fill_holes is scipy.ndimage.binary_fill_holes, as nnU-Net's create_nonzero_mask calls it [not in tree].
Pinned: tests/golden/rows_ref.npz holds the outputs of the REFERENCE's own remove_sparks / remove_3mm_sparks / connected_components
bodies (executed from /root/reference in the build container by tests/golden/make_golden_ref.py); tests/test_rows_ref.py holds
these functions to them bit for bit.  fill_holes (nnU-Net, not in tree) stays unpinned.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this.
"""
import numpy as np
from scipy.ndimage import binary_fill_holes, label


def remove_sparks(mask, min_volume=3):
    mask = (np.asarray(mask) > 0.5).astype('int')
    labeled, n = label(mask)
    sizes = np.bincount(labeled.ravel(), minlength=n + 1)
    keep = sizes >= min_volume
    keep[0] = False
    return keep[labeled].astype('int')


def remove_3mm_sparks(mask, voxel_size):
    if not isinstance(voxel_size, list):
        raise RuntimeError('voxel_size should be a list of 3 floats.')
    voxel_volume = voxel_size[0] * voxel_size[1] * voxel_size[2]
    voxel_aniso = np.max(voxel_size) / np.min(voxel_size)
    if voxel_aniso > 3.0:
        return remove_sparks(mask, min_volume=3)
    min_volume = int(np.around(3.0 / voxel_volume))
    if min_volume < 2:
        min_volume = 2
    return remove_sparks(mask, min_volume=min_volume)


def fill_holes(mask):
    return binary_fill_holes(np.asarray(mask) != 0)


def components_partition(mask):
    """Canonical labelling for comparing partitions: every voxel carries the smallest flat index of its component."""
    mask = np.asarray(mask) != 0
    labeled, n = label(mask)
    flat = labeled.ravel()
    first = np.full(n + 1, -1, dtype=np.int64)
    idx = np.arange(flat.size, dtype=np.int64)
    order = np.argsort(flat, kind='stable')
    sl = flat[order]
    starts = np.concatenate(([True], sl[1:] != sl[:-1]))
    first[sl[starts]] = idx[order][starts]
    out = first[flat]
    out[flat == 0] = -1
    return out.reshape(mask.shape)
