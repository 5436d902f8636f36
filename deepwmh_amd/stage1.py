"""DeepWMH Stage I -- negative-log-likelihood initial segmentation -- on the GPU.

Mirrors /root/reference/deepwmh/analysis/lesion_analysis.py (nll :84-113, nll_analysis :115-278, LesionAnalyzer
:358-460) and the helpers it takes from image_ops.py (z_score :172-179, mean_std_grid :56-170, component_filtering
:256-308, median_3mm :379-421, average_contiguous_labels :23-38), with the reference's names and argument meaning.
All voxel work runs in libdeepwmh_amd.so in double (the reference is float64 numpy); what stays on the host is the
256- / 400-bin histogram post-processing (Otsu's between-class variance, zero crossings) on exact integer counts.
Volumes stay on the device between the steps of `nll_analysis_arrays`.  No CPU fallback.
"""
import os
import pickle
import shutil

import numpy as np

from . import _lib, nifti
from .postprocess import remove_3mm_sparks

MIN_STD = 0.03           # lesion_analysis.py:180


def _torch():
    import torch
    return torch


def _f32(a):
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device='cuda', dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _f64(a):
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device='cuda', dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def _u8(mask):
    torch = _torch()
    if isinstance(mask, torch.Tensor):
        return (mask if mask.dtype == torch.uint8 else (mask > 0.5).to(torch.uint8)).cuda().contiguous()
    return torch.from_numpy(np.ascontiguousarray((np.asarray(mask) > 0.5).astype(np.uint8))).cuda()


def _iarr(v):
    return _lib.iarr(v)


class _Ctx:
    """Library handle, stream and the small reusable device scratch."""
    def __init__(self):
        torch = _torch()
        self.lib = _lib.load()
        self.ws = torch.empty(self.lib.wmh_moments_workspace_doubles(), dtype=torch.float64, device='cuda')

    @property
    def st(self):
        return _lib.current_stream()

    def moments(self, x, mask=None, shift=0.0):
        """device tensor of 7 doubles: count, mean, std, min in mask, min all, max in mask, max all"""
        torch = _torch()
        out = torch.empty(7, dtype=torch.float64, device='cuda')
        fn = self.lib.wmh_masked_moments_f32 if x.dtype == torch.float32 else self.lib.wmh_masked_moments_f64
        _lib.check(fn(_lib.ptr(x), _lib.ptr(mask), x.numel(), float(shift), _lib.ptr(out), _lib.ptr(self.ws), self.st), 'wmh_masked_moments')
        return out


# ---- primitives with the reference's signatures (numpy in, numpy out) -------------------------------------------------
def z_score(data, mask=None):
    """(data - mean) / max(std, 1e-5), statistics inside mask > 0.5 when given (image_ops.py:172-179).  float64 out."""
    torch = _torch()
    c = _Ctx()
    x = _f32(data)
    m = _u8(mask) if mask is not None else None
    out = _zscore_dev(c, x, m, None)
    return out.cpu().numpy().reshape(np.asarray(data).shape)


def _zscore_dev(c, x32, mask_u8, fill_mask):
    """z-score of a float32 device volume; with fill_mask the outside becomes the minimum inside it (tissue_min)."""
    torch = _torch()
    rough = c.moments(x32, mask_u8, 0.0)
    mean0 = float(rough[1])
    mom = c.moments(x32, mask_u8, mean0).cpu().tolist()                  # second pass about the mean: no cancellation
    mean, std = mom[1], max(mom[2], 0.00001)
    out = torch.empty(x32.shape, dtype=torch.float64, device='cuda')
    _lib.check(c.lib.wmh_zscore_f64(_lib.ptr(x32), mean, std, None, None, _lib.ptr(out), x32.numel(), c.st), 'wmh_zscore_f64')
    if fill_mask is not None:
        zm = c.moments(out, fill_mask, 0.0)
        _lib.check(c.lib.wmh_mask_fill_f64(_lib.ptr(out), _lib.ptr(fill_mask), _lib.ptr(zm[3:4]), out.numel(), c.st), 'wmh_mask_fill_f64')
    return out


def _histogram_dev(c, x64, edges, mask=None, replace_mask=None, replace_value=None):
    torch = _torch()
    e = torch.from_numpy(np.ascontiguousarray(edges, dtype=np.float64)).cuda()
    counts = torch.empty(len(edges) - 1, dtype=torch.int64, device='cuda')
    _lib.check(c.lib.wmh_histogram_f64(_lib.ptr(x64), _lib.ptr(mask), _lib.ptr(replace_mask), _lib.ptr(replace_value), x64.numel(), _lib.ptr(e),
                                       len(edges) - 1, _lib.ptr(counts), c.st), 'wmh_histogram_f64')
    return counts.cpu().numpy()


def _otsu_from_counts(counts, edges):
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype('float32', copy=False)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[np.argmax(var12)]


def threshold_otsu(image, nbins=256):
    """scikit-image's threshold_otsu (the reference's dependency): histogram on the GPU, 256-bin search on the host."""
    c = _Ctx()
    x = _f64(image).reshape(-1)
    mom = c.moments(x, None, 0.0).cpu().tolist()
    lo, hi = mom[4], mom[6]
    if lo == hi:
        return np.float64(lo)
    edges = np.linspace(lo, hi, nbins + 1)
    return _otsu_from_counts(_histogram_dev(c, x, edges), edges)


def _grid_geometry(shape, patch_size):
    patch = [int(v) for v in 2 * np.ceil(np.array(patch_size) / 2).astype('int')]
    step = [p // 2 for p in patch]
    pshape = [int(v) for v in np.array(patch) * np.ceil(np.array(shape) / np.array(patch)).astype('int')]
    gshape = [pshape[a] // step[a] for a in range(3)]
    zoomed = [int(round((g + 2) * s)) for g, s in zip(gshape, step)]
    offset = [s // 2 for s in step]
    return patch, step, gshape, zoomed, offset


def _mean_std_grid_dev(c, x64, patch_size, mask_u8, want_std=False):
    torch = _torch()
    D, H, W = (int(v) for v in x64.shape)
    patch, step, gshape, zoomed, offset = _grid_geometry((D, H, W), patch_size)
    cells = int(np.prod(gshape))
    mg = torch.empty(cells, dtype=torch.float64, device='cuda')
    sg = torch.empty(cells, dtype=torch.float64, device='cuda')
    _lib.check(c.lib.wmh_block_stats(_lib.ptr(x64), _lib.ptr(mask_u8), D, H, W, _iarr(patch), _iarr(step), _iarr(gshape), _lib.ptr(mg),
                                     _lib.ptr(sg), c.st), 'wmh_block_stats')
    outs = []
    for g in ((mg, sg) if want_std else (mg,)):
        padded = torch.zeros([v + 2 for v in gshape], dtype=torch.float64, device='cuda')
        padded[1:-1, 1:-1, 1:-1] = g.view(gshape)
        out = torch.empty((D, H, W), dtype=torch.float64, device='cuda')
        _lib.check(c.lib.wmh_zoom_linear_window(_lib.ptr(padded.contiguous()), _iarr([v + 2 for v in gshape]), _iarr(zoomed), _iarr(offset),
                                                D, H, W, _lib.ptr(out), c.st), 'wmh_zoom_linear_window')
        outs.append(out)
    return outs if want_std else outs[0]


def mean_std_grid(data, patch_size, order=1, mask=None):
    """Coarse local mean / std fields (image_ops.py:56-170); order must be 1 (the only value the reference uses)."""
    if order != 1:
        raise NotImplementedError('mean_std_grid: order 1 is implemented')
    c = _Ctx()
    m, s = _mean_std_grid_dev(c, _f64(data), patch_size, _u8(mask) if mask is not None else None, want_std=True)
    return m.cpu().numpy(), s.cpu().numpy()


def _side(side):
    assert side in [None, '+', '-']
    return 0 if side is None else (1 if side == '+' else -1)


def nll(x_prime, x_refs, min_std=None, side=None, return_all=False, use_mask=False):
    """Anomaly score = NLL under the per-voxel Gaussian fitted to the references (lesion_analysis.py:84-113).  use_mask: every
    reference counts only above its own Otsu threshold (:87-92; not used by the reference's pipeline)."""
    torch = _torch()
    c = _Ctx()
    xp = _f64(x_prime)
    refs = torch.stack([_f64(r) for r in x_refs])
    shape = np.asarray(x_prime).shape
    if use_mask:
        thr = torch.tensor([float(threshold_otsu(r)) for r in x_refs], dtype=torch.float64, device='cuda')
        anomaly, mu, sigma = torch.empty_like(xp), torch.empty_like(xp), torch.empty_like(xp)
        _lib.check(c.lib.wmh_nll_masked(_lib.ptr(xp), _lib.ptr(refs), _lib.ptr(thr), int(refs.shape[0]), xp.numel(),
                                        -1.0 if min_std is None else float(min_std), _side(side), _lib.ptr(anomaly), _lib.ptr(mu), _lib.ptr(sigma),
                                        c.st), 'wmh_nll_masked')
    else:
        anomaly, mu, sigma, _ = _nll_dev(c, xp, refs, min_std, side, with_refs=False)
    if not return_all:
        return anomaly.cpu().numpy().reshape(shape)
    return anomaly.cpu().numpy().reshape(shape), mu.cpu().numpy().reshape(shape), sigma.cpu().numpy().reshape(shape)


def _nll_dev(c, xp, refs, min_std, side, with_refs):
    torch = _torch()
    n, R = xp.numel(), int(refs.shape[0])
    anomaly = torch.empty_like(xp); mu = torch.empty_like(xp); sigma = torch.empty_like(xp)
    arefs = torch.empty_like(refs) if with_refs else None
    _lib.check(c.lib.wmh_nll(_lib.ptr(xp), _lib.ptr(refs), R, n, -1.0 if min_std is None else float(min_std), _side(side), _lib.ptr(anomaly),
                             _lib.ptr(mu), _lib.ptr(sigma), _lib.ptr(arefs), c.st), 'wmh_nll')
    return anomaly, mu, sigma, arefs


def _component_filtering_dev(c, mask_u8, voxel_size):
    torch = _torch()
    D, H, W = (int(v) for v in mask_u8.shape)
    do = [0, 0, 0]
    if np.max(voxel_size) / np.min(voxel_size) > 3:
        do[int(np.argmax(voxel_size))] = 1
    else:
        do = [1, 1, 1]
    n = D * H * W
    out = torch.empty_like(mask_u8)
    scratch = torch.empty_like(mask_u8)
    labels = torch.empty(n, dtype=torch.int32, device='cuda'); counts = torch.empty_like(labels)
    best = torch.empty(max(D, H, W), dtype=torch.int64, device='cuda')
    _lib.check(c.lib.wmh_component_filtering(_lib.ptr(mask_u8), _lib.ptr(out), _lib.ptr(scratch), _lib.ptr(labels), _lib.ptr(counts), _lib.ptr(best),
                                             D, H, W, _iarr(do), c.st), 'wmh_component_filtering')
    return out


def component_filtering(mask, voxel_size, return_type='float32', erosion=True):
    """Per slice and orientation: cross erosion, then the largest component (image_ops.py:256-308).  `erosion` is accepted and,
    as in the reference, has no effect: the reference's body erodes unconditionally (image_ops.py:286,293,300 never read it)."""
    return _component_filtering_dev(_Ctx(), _u8(mask), voxel_size).cpu().numpy().astype(return_type)


def _median_kernel(voxel_size):
    maxl, minl = np.max(voxel_size), np.min(voxel_size)
    if maxl / minl > 4.0:
        ax = int(np.argmax(voxel_size))
        ks = [max(int(3.0 / voxel_size[a]), 3) for a in range(3) if a != ax]
        k3 = ks[:]
        k3.insert(ax, 1)                                           # per-slice 2-D filter = 3-D window of extent 1 on that axis
        return k3
    return [max(int(3.0 / r), 3) for r in voxel_size]


def _median_dev(c, x64, voxel_size):
    torch = _torch()
    D, H, W = (int(v) for v in x64.shape)
    k = _median_kernel(voxel_size)
    out = torch.empty_like(x64)
    _lib.check(c.lib.wmh_median_filter_f64(_lib.ptr(x64), _lib.ptr(out), D, H, W, k[0], k[1], k[2], c.st), 'wmh_median_filter_f64')
    return out


def median_3mm(data, physical_voxel_size):
    """3 mm median smoothing (image_ops.py:379-421): 3-D window, or per-slice 2-D for thick-slice data."""
    return _median_dev(_Ctx(), _f64(data), physical_voxel_size).cpu().numpy()


def average_contiguous_labels(labels):
    """Per-voxel majority label over label maps with contiguous ids (image_ops.py:23-38)."""
    torch = _torch()
    c = _Ctx()
    stack = torch.stack([_f32(l) for l in labels])
    K = int(max(int(np.max(np.asarray(l))) for l in labels)) + 1
    out = torch.empty(stack.shape[1:], dtype=torch.uint8, device='cuda')
    _lib.check(c.lib.wmh_label_vote(_lib.ptr(stack), len(labels), K, _lib.ptr(out), out.numel(), c.st), 'wmh_label_vote')
    return out.cpu().numpy().astype(np.int64)


def _log_curve(counts):
    hist = np.where(counts == 0, 0.001, counts)
    hist = np.log10(hist)
    return np.where(hist < 0, 0, hist)


# ---- the analysis (lesion_analysis.py:115-278) on arrays ----------------------------------------------------------------
def nll_analysis_arrays(x_input, x_refs, label1s, label2s, voxel_size, intensity_prior=None, apply_otsu=True, mean_correction=True):
    """x_input / x_refs: float32 volumes (registered); label1s: brain masks of the references; label2s: tissue labels
    (0 background, 1 cerebrum, 2 cerebellum + brainstem, 3 cortex).  Returns a dict of numpy results: anomaly,
    valid_mask, normalized_input, curve_x / curve_y / curve_r, threshold, segmentation."""
    torch = _torch()
    assert intensity_prior in [None, '+', '-'], 'Unknown intensity prior "%s".' % str(intensity_prior)
    c = _Ctx()
    lib, st = c.lib, c.st
    shape = tuple(np.asarray(x_input).shape)
    n, R = int(np.prod(shape)), len(x_refs)
    patch = [int(v) for v in np.ceil([50 / voxel_size[0], 50 / voxel_size[1], 50 / voxel_size[2]]).astype('int')]

    l1 = torch.stack([_f32(m) for m in label1s])
    rough = torch.empty(shape, dtype=torch.uint8, device='cuda')
    _lib.check(lib.wmh_vote_mask(_lib.ptr(l1), len(label1s), _lib.ptr(rough), n, st), 'wmh_vote_mask')
    del l1

    x_prime = _zscore_dev(c, _f32(x_input), rough, None)
    if apply_otsu:
        mom = c.moments(x_prime, rough, 0.0)                       # [4] = x_prime.min(), [5] = max inside the brain
        lo, hi_in = float(mom[4]), float(mom[5])
        edges = np.linspace(lo, max(lo, hi_in), 257)
        counts = _histogram_dev(c, x_prime.view(-1), edges, None, rough.view(-1), mom[4:5])
        otsu_thr = np.float64(lo) if lo == max(lo, hi_in) else _otsu_from_counts(counts, edges)
        valid = ((x_prime > float(otsu_thr)) & (rough != 0)).to(torch.uint8)
    else:
        otsu_thr = None
        valid = rough.clone()
    zm = c.moments(x_prime, rough, 0.0)
    _lib.check(lib.wmh_mask_fill_f64(_lib.ptr(x_prime), _lib.ptr(rough), _lib.ptr(zm[3:4]), n, st), 'wmh_mask_fill_f64')

    refs = torch.empty((R,) + shape, dtype=torch.float64, device='cuda')
    for i, t in enumerate(x_refs):
        refs[i] = _zscore_dev(c, _f32(t), rough, rough)
    mu_p = _mean_std_grid_dev(c, x_prime, patch, valid)
    if mean_correction:
        for i in range(R):
            mu_i = _mean_std_grid_dev(c, refs[i], patch, valid)
            _lib.check(lib.wmh_sub_add_f64(_lib.ptr(refs[i]), _lib.ptr(mu_i), _lib.ptr(mu_p), n, st), 'wmh_sub_add_f64')

    anomaly, x_mean, x_std, arefs = _nll_dev(c, x_prime, refs, MIN_STD, intensity_prior, with_refs=True)
    filt = _component_filtering_dev(c, valid, voxel_size)
    _lib.check(lib.wmh_mul_mask_f64(_lib.ptr(anomaly), _lib.ptr(filt), n, st), 'wmh_mul_mask_f64')
    for i in range(R):
        _lib.check(lib.wmh_mul_mask_f64(_lib.ptr(arefs[i]), _lib.ptr(valid), n, st), 'wmh_mul_mask_f64')

    # histogram curves (lesion_analysis.py:52-82): bins from the references' mean positive score inside the valid mask
    ref_means = []
    for i in range(R):
        pos = ((arefs[i] > 0) & (valid != 0)).to(torch.uint8)
        ref_means.append(float(c.moments(arefs[i], pos, 0.0)[1]))
    bin_width = np.array(ref_means).mean() / 4
    bins = np.linspace(0.0, 0.0 + 400 * bin_width, num=401)
    centers = np.array([(bins[i] + bins[i + 1]) / 2 for i in range(400)])
    curve_y = _log_curve(_histogram_dev(c, anomaly.view(-1), bins))
    curve_rs = [_log_curve(_histogram_dev(c, arefs[i].view(-1), bins)) for i in range(R)]
    curve_r = np.zeros_like(centers)
    for r0 in curve_rs:
        curve_r += r0
    curve_r = curve_r / R
    crossings = []
    for r0 in curve_rs:
        for j in range(len(r0) - 1, 0, -1):
            if r0[j] > 0.01:
                crossings.append(centers[j])
                break
    threshold = np.median(np.sort(crossings))
    del arefs

    # tissue priors (lesion_analysis.py:217-243)
    l2 = torch.stack([_f32(t) for t in label2s])
    K = int(max(int(np.max(np.asarray(t))) for t in label2s)) + 1
    averaged = torch.empty(shape, dtype=torch.uint8, device='cuda')
    _lib.check(lib.wmh_label_vote(_lib.ptr(l2), len(label2s), K, _lib.ptr(averaged), n, st), 'wmh_label_vote')
    _lib.check(lib.wmh_mul_mask_f64(_lib.ptr(anomaly), _lib.ptr((averaged > 0).to(torch.uint8)), n, st), 'wmh_mul_mask_f64')
    med = _median_dev(c, anomaly, voxel_size)
    _lib.check(lib.wmh_select_f64(_lib.ptr(anomaly), _lib.ptr(med), _lib.ptr((averaged == 2).to(torch.uint8)), n, st), 'wmh_select_f64')
    tissue = torch.empty(shape, dtype=torch.uint8, device='cuda')
    if len(label2s) != R:
        raise ValueError('nll_analysis: one tissue label per reference expected')
    _lib.check(lib.wmh_vote_mask(_lib.ptr(l2), R, _lib.ptr(tissue), n, st), 'wmh_vote_mask')
    _lib.check(lib.wmh_mul_mask_f64(_lib.ptr(anomaly), _lib.ptr(tissue), n, st), 'wmh_mul_mask_f64')

    seg = (anomaly > float(threshold)).to(torch.float32)
    return dict(anomaly=anomaly.cpu().numpy(), valid_mask=valid.cpu().numpy().astype('int'), normalized_input=x_prime.cpu().numpy(),
                curve_x=centers, curve_y=curve_y, curve_r=curve_r, threshold=threshold, otsu_threshold=otsu_thr,
                x_mean=x_mean.cpu().numpy(), x_std=x_std.cpu().numpy(), rough_brain=rough.cpu().numpy().astype('int'),
                segmentation=seg.cpu().numpy())


def _pixdim(meta):
    return [float(v) for v in meta['spacing']]


def nll_analysis(case_info, apply_otsu=True, intensity_prior=None, case_output_folder=None, mean_correction=True, debug=False):
    """File-level form with the reference's signature and outputs (lesion_analysis.py:115-278): reads case_info
    {'x', 'r', 'm', 'y'} paths, writes normalized_input / anomaly_score / valid_mask .nii.gz into case_output_folder and
    returns (anomaly, m_valid_score, curve_x, curve_y, curve_r, anomaly_threshold).  The histogram PNG and the debug
    dumps are not produced."""
    assert intensity_prior in [None, '+', '-'], 'Unknown intensity prior "%s".' % str(intensity_prior)
    assert case_output_folder is not None, 'must assign case output folder.'
    if debug:
        raise NotImplementedError('nll_analysis: debug dumps are not implemented')
    x, meta = nifti.read(case_info['x'])
    refs = [nifti.read(p)[0] for p in case_info['r']]
    l1 = [nifti.read(p)[0] for p in case_info['m']]
    l2 = [nifti.read(p)[0] for p in case_info['y']]
    res = nll_analysis_arrays(x, refs, l1, l2, _pixdim(meta), intensity_prior, apply_otsu, mean_correction)
    os.makedirs(case_output_folder, exist_ok=True)
    nifti.write(os.path.join(case_output_folder, 'normalized_input.nii.gz'), res['normalized_input'], like=meta, dtype=np.float32)
    nifti.write(os.path.join(case_output_folder, 'anomaly_score.nii.gz'), res['anomaly'], like=meta, dtype=np.float32)
    nifti.write(os.path.join(case_output_folder, 'valid_mask.nii.gz'), res['valid_mask'], like=meta, dtype=np.float32)
    return res['anomaly'], res['valid_mask'], res['curve_x'], res['curve_y'], res['curve_r'], res['threshold']


class LesionAnalyzer(object):
    """Same surface as the reference class (lesion_analysis.py:358-460): add_case, analyze_and_do_segmentation,
    do_postprocessing; cases run one after another on the GPU instead of in a process pool."""

    def __init__(self, output_folder, num_workers=8, logger=None):
        self.data_dict = {}
        self.output_folder = output_folder
        os.makedirs(output_folder, exist_ok=True)
        self.normalization_method = 'z_score'
        self.apply_otsu = True
        self.num_workers = num_workers
        self.logger = logger
        self.debug = False

    def log(self, msg, print_to_console=True):
        if self.logger is not None:
            self.logger.write(msg, timestamp=True)
        if print_to_console:
            print(msg)

    def add_case(self, name, x_input, x_refs, label1, label2):
        self.data_dict[name] = {'x': x_input, 'r': x_refs, 'm': label1, 'y': label2}

    def analyze_and_do_segmentation(self, intensity_prior=None, do_postprocessing=True):
        assert intensity_prior in [None, '+', '-']
        for case, info in self.data_dict.items():
            folder = os.path.join(self.output_folder, case)
            os.makedirs(folder, exist_ok=True)
            summary_path = os.path.join(folder, 'summary.pkl')
            if not os.path.isfile(summary_path):
                _, _, cx, cy, cr, thr = nll_analysis(info, apply_otsu=self.apply_otsu, intensity_prior=intensity_prior,
                                                     case_output_folder=folder, debug=self.debug)
                shutil.copyfile(info['x'], os.path.join(folder, 'preprocessed_image.nii.gz'))
                summary = {'preprocessed_image': info['x'], 'analyzer_name': type(self).__name__,
                           'normalization_method': self.normalization_method, 'apply_otsu': self.apply_otsu, 'output_folder': folder,
                           'anomaly_score': os.path.join(folder, 'anomaly_score.nii.gz'),
                           'otsu_mask': os.path.join(folder, 'otsu_mask.nii.gz') if self.apply_otsu else None,
                           'histogram_curves': {'x': cx, 'y': cy, 'r': cr}, 'autoseg_threshold': thr}
                with open(summary_path, 'wb') as f:
                    pickle.dump(summary, f)
            seg_path = os.path.join(folder, 'segmentation.nii.gz')
            if not os.path.isfile(seg_path):                          # _parallel_segmentation, :332-353
                with open(summary_path, 'rb') as f:
                    summary = pickle.load(f)
                score, meta = nifti.read(summary['anomaly_score'])
                thr = summary['autoseg_threshold']
                nifti.write(seg_path, (score > thr).astype('float32'), like=meta, dtype=np.float32)
                summary['final_threshold'] = thr
                summary['segmentation_file'] = seg_path
                with open(summary_path, 'wb') as f:
                    pickle.dump(summary, f)
                with open(os.path.join(folder, 'segmentation.txt'), 'w') as f:
                    f.write('case name: %s\n' % case)
                    f.write('segmentation threshold: %.4f\n' % thr)
        if do_postprocessing:
            self.do_postprocessing()

    def do_postprocessing(self):
        for case in self.data_dict:                                   # _parallel_post_processing, :320-330
            folder = os.path.join(self.output_folder, case)
            out = os.path.join(folder, 'segmentation_pp.nii.gz')
            if os.path.isfile(out):
                continue
            seg, meta = nifti.read(os.path.join(folder, 'segmentation.nii.gz'))
            pp = remove_3mm_sparks(seg, _pixdim(meta))
            nifti.write(out, pp.astype(np.float32), like=meta, dtype=np.float32)
