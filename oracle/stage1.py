"""TEST INFRASTRUCTURE ONLY -- numpy/scipy restatement of DeepWMH's Stage I (NLL initial segmentation).

Follows, operation for operation and dtype for dtype (float32 inputs as load_nifti returns them, float64 after the
z-score), the reference functions:
  masked_mean / masked_std / z_score          /root/reference/deepwmh/analysis/image_ops.py:13-21,172-179
  mean_std_grid                               image_ops.py:56-170
  group_mean / group_std                      image_ops.py:197-231
  component_filtering                         image_ops.py:256-308
  median_3mm                                  image_ops.py:379-421
  average_contiguous_labels                   image_ops.py:23-38
  nll, hist_curve, histogram_analysis, nll_analysis   /root/reference/deepwmh/analysis/lesion_analysis.py:40-113,115-278
threshold_otsu is scikit-image's (an unpinned dependency of the reference, absent here): restated from its published
algorithm (256-bin histogram over [min, max], float32 counts, maximum between-class variance, returns the bin centre).
Array-in / array-out (the reference reads NIfTI paths).  Pinned: every function above except threshold_otsu and the nll_analysis
driver is held bit for bit to the outputs of the REFERENCE's own function bodies (tests/golden/rows_ref.npz, made by
tests/golden/make_golden_ref.py from /root/reference; tests/test_rows_ref.py).  threshold_otsu (scikit-image, absent) and therefore
nll(use_mask=True) and the composition in nll_analysis_arrays stay unpinned.  Only tests/, smoke() and bench.py's cpu_baseline leg may import this.
"""
import warnings

import numpy as np
import numpy.ma as ma
import scipy.ndimage
from scipy.ndimage import binary_erosion, label, zoom


def masked_mean(data, mask):
    mask = (mask > 0.5).astype('int')
    return ma.masked_array(data, mask=1 - mask).mean()


def masked_std(data, mask):
    mask = (mask > 0.5).astype('int')
    return ma.masked_array(data, mask=1 - mask).std()


def z_score(data, mask=None):
    data_mean = np.mean(data) if mask is None else masked_mean(data, mask)
    data_std = np.std(data) if mask is None else masked_std(data, mask)
    data_std = np.max([data_std, 0.00001])
    return (data - data_mean) / data_std


def threshold_otsu(image, nbins=256):
    flat = np.asarray(image).reshape(-1)
    if np.all(flat == flat[0]):
        return flat[0]
    counts, edges = np.histogram(flat, bins=nbins, range=None)
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype('float32', copy=False)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[np.argmax(var12)]


def mean_std_grid(data, patch_size, order=1, mask=None):
    patch_size = list(2 * np.ceil(np.array(patch_size) / 2).astype('int'))
    step = [p // 2 for p in patch_size]
    shp = data.shape
    pshape = list(np.array(patch_size) * np.ceil(np.array(shp) / np.array(patch_size)).astype('int'))
    pdata = np.zeros(pshape).astype('float32')
    pdata[:shp[0], :shp[1], :shp[2]] = data
    if mask is not None:
        pmask = np.zeros_like(pdata)
        pmask[:shp[0], :shp[1], :shp[2]] = (mask > 0.5).astype('float32')
    gshape = [pshape[a] // step[a] for a in range(3)]
    mean_grid, std_grid = np.zeros(gshape), np.zeros(gshape)
    for i in range(gshape[0]):
        for j in range(gshape[1]):
            for k in range(gshape[2]):
                sl = tuple(slice(c * s, c * s + p) for c, s, p in zip((i, j, k), step, patch_size))
                block = pdata[sl]
                if mask is not None:
                    mblock = pmask[sl]
                    if np.sum(mblock) > 0:
                        bm = ma.masked_array(block, mask=1 - mblock)
                        mu, sigma = bm.mean(), bm.std()
                    else:
                        mu, sigma = 0, 0.00001
                else:
                    mu, sigma = np.mean(block), np.max([np.std(block), 0.00001])
                mean_grid[i, j, k], std_grid[i, j, k] = mu, sigma
    outs = []
    for grid in (mean_grid, std_grid):
        g = np.zeros([s + 2 for s in gshape])
        g[1:-1, 1:-1, 1:-1] = grid
        z = zoom(g, step, order=order)
        off = [s // 2 for s in step]
        z = z[off[0]:off[0] + gshape[0] * step[0], off[1]:off[1] + gshape[1] * step[1], off[2]:off[2] + gshape[2] * step[2]]
        outs.append(z[:shp[0], :shp[1], :shp[2]])
    return outs[0], outs[1]


def _group(fn, data_list, masks):
    if masks is None:
        masks = [None] * len(data_list)
    rows, shape = [], None
    for d, m in zip(data_list, masks):
        if m is not None:
            d = np.where(m < 0.5, np.nan, d)
        shape = d.shape
        rows.append(np.reshape(d, [-1]))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return np.reshape(fn(np.vstack(rows), axis=0), shape)


def group_mean(data_list, masks=None):
    return _group(np.nanmean, data_list, masks)


def group_std(data_list, masks=None):
    return _group(np.nanstd, data_list, masks)


def _largest_component(mask2d, return_type='float32'):
    lab, nf = label((mask2d > 0.5).astype('int'))
    if nf == 0:
        return np.zeros_like(lab).astype(return_type)
    sizes = np.bincount(lab.ravel(), minlength=nf + 1)[1:]
    return (lab == int(np.argmax(sizes)) + 1).astype(return_type)          # first of the largest, as the `<` scan does


def component_filtering(mask, voxel_size, return_type='float32'):
    do = [False] * 3
    if np.max(voxel_size) / np.min(voxel_size) > 3:
        do[int(np.argmax(voxel_size))] = True
    else:
        do = [True] * 3
    vols = []
    for axis in range(3):
        v = np.zeros_like(mask)
        for s in range(mask.shape[axis]):
            idx = [slice(None)] * 3
            idx[axis] = s
            idx = tuple(idx)
            v[idx] = _largest_component(binary_erosion(mask[idx])) if do[axis] else mask[idx]
        vols.append(v)
    return ((vols[2] + vols[1] + vols[0]) > 0.5).astype(return_type)


def median_kernel_size(physical_voxel_size):
    """(axis or None, kernel) exactly as median_3mm derives it."""
    maxl, minl = np.max(physical_voxel_size), np.min(physical_voxel_size)
    if maxl / minl > 4.0:
        ax = int(np.argmax(physical_voxel_size))
        res = [physical_voxel_size[a] for a in range(3) if a != ax]
        ks = [max(int(3.0 / r), 3) for r in res]
        return ax, ks
    return None, [max(int(3.0 / r), 3) for r in physical_voxel_size]


def median_3mm(data, physical_voxel_size):
    ax, ks = median_kernel_size(physical_voxel_size)
    if ax is None:
        return scipy.ndimage.median_filter(data, size=ks, mode='constant', cval=0)
    out = np.zeros(data.shape)
    for s in range(data.shape[ax]):
        idx = [slice(None)] * 3
        idx[ax] = s
        idx = tuple(idx)
        out[idx] = scipy.ndimage.median_filter(data[idx], size=ks, mode='constant', cval=0)
    return out


def average_contiguous_labels(labels):
    nch = 0
    for lab in labels:
        nch = max(nch, int(np.max(lab)) + 1)
    channels = np.zeros([nch, *labels[0].shape])
    for lab in labels:
        il = lab.astype('int')
        for ch in range(nch):
            channels[ch] += (il == ch).astype('float32')
    return np.argmax(channels, axis=0)


def nll(x_prime, x_refs, min_std=None, side=None, return_all=False, use_mask=False):
    if use_mask:                                  # lesion_analysis.py:87-92
        masks = [np.where(x_i > threshold_otsu(x_i), 1, 0) for x_i in x_refs]
        mu, sigma = group_mean(x_refs, masks=masks), group_std(x_refs, masks=masks)
    else:
        mu, sigma = group_mean(x_refs), group_std(x_refs)
    if min_std is None:
        sigma += 1e-6
    else:
        sigma = np.where(sigma < min_std, min_std, sigma)
    anomaly = np.power(x_prime - mu, 2) / (2 * np.power(sigma, 2)) + np.log(sigma * 2.506)
    anomaly = np.nan_to_num(anomaly, nan=0.0)
    if side == '+':
        anomaly = anomaly * (x_prime > mu).astype('float32')
    elif side == '-':
        anomaly = anomaly * (x_prime < mu).astype('float32')
    return (anomaly, mu, sigma) if return_all else anomaly


def hist_curve(data, bins, log_y=False, mask=None):
    hist, edges = np.histogram(data if mask is None else data[mask > 0.5], bins=bins)
    centers = np.array([(edges[i] + edges[i + 1]) / 2 for i in range(len(hist))])
    if log_y:
        hist = np.where(hist == 0, 0.001, hist)
        hist = np.log10(hist)
        hist = np.where(hist < 0, 0, hist)
    return centers, hist


def histogram_analysis(a_prime, a_refs, mask):
    ref_means = []
    for a in a_refs:
        f = a[mask > 0.5]
        f = f[f > 0]
        ref_means.append(f.mean())
    bin_width = np.array(ref_means).mean() / 4
    num_bins = 400
    bins = np.linspace(0.0, 0.0 + num_bins * bin_width, num=num_bins + 1)
    x, y = hist_curve(a_prime, bins, log_y=True)
    r, rs = np.zeros_like(x), []
    for a in a_refs:
        _, r0 = hist_curve(a, bins, log_y=True)
        r += r0
        rs.append(r0)
    return x, y, r / len(a_refs), rs, bins


def nll_analysis_arrays(x_input, x_refs, label1s, label2s, voxel_size, intensity_prior=None, apply_otsu=True, mean_correction=True):
    """nll_analysis (lesion_analysis.py:115-278) on arrays: x_input / x_refs float32 volumes, label1s (brain) and label2s
    (tissue ids 0..3) float32.  Returns a dict with the fields the reference saves or returns."""
    patch = list(np.ceil([50 / voxel_size[0], 50 / voxel_size[1], 50 / voxel_size[2]]).astype('int'))
    m_i = [(m > 0.5).astype('float32') for m in label1s]
    m_rough = (group_mean(m_i) > 0.5).astype('int')
    x_prime = z_score(x_input, mask=m_rough)
    if apply_otsu:
        thr = threshold_otsu(np.where(m_rough < 0.5, x_prime.min(), x_prime))
        m_otsu = np.where(x_prime > thr, 1, 0)
    else:
        thr, m_otsu = None, np.ones_like(x_prime)
    m_valid = m_rough * m_otsu
    tissue_min = ma.masked_array(x_prime, mask=1 - m_rough).min()
    x_prime = np.where(m_rough < 0.5, tissue_min, x_prime)
    x_i = []
    for t in x_refs:
        t = z_score(t, mask=m_rough)
        tmin = ma.masked_array(t, mask=1 - m_rough).min()
        x_i.append(np.where(m_rough < 0.5, tmin, t))
    mu_p, _ = mean_std_grid(x_prime, patch, mask=m_valid)
    if mean_correction:
        for i in range(len(x_i)):
            mu_i, _ = mean_std_grid(x_i[i], patch, mask=m_valid)
            x_i[i] = x_i[i] - mu_i + mu_p
    min_std = 0.03
    anomaly, x_mean, x_std = nll(x_prime, x_i, min_std=min_std, side=intensity_prior, return_all=True)
    anomaly = anomaly * component_filtering(m_valid, voxel_size)
    anomaly_refs = [nll(s, x_i, min_std=min_std, side=intensity_prior) * m_valid for s in x_i]
    cx, cy, cr, crs, bins = histogram_analysis(anomaly, anomaly_refs, mask=m_valid)
    crossings = []
    for c in crs:
        for j in range(len(c) - 1, 0, -1):
            if c[j] > 0.01:
                crossings.append(cx[j])
                break
    threshold = np.median(np.sort(crossings))
    averaged = average_contiguous_labels(label2s)
    anomaly = anomaly * (averaged > 0.5).astype('float32')
    cb = ((1.5 < averaged) * (averaged < 2.5)).astype('float32')
    anomaly = np.where(cb > 0.5, median_3mm(anomaly, voxel_size), anomaly)
    tissue = np.zeros(m_valid.shape).astype('float32')
    for t in label2s:
        tissue += (t > 0.5).astype('float32')
    tissue = (tissue > (len(x_refs) / 2)).astype('float32')
    anomaly = anomaly * tissue
    return dict(anomaly=anomaly, valid_mask=m_valid, normalized_input=x_prime, curve_x=cx, curve_y=cy, curve_r=cr,
                threshold=threshold, otsu_threshold=thr, x_mean=x_mean, x_std=x_std, rough_brain=m_rough,
                segmentation=(anomaly > threshold).astype('float32'))
