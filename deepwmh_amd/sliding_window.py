"""Sliding-window Gaussian-weighted patch inference on the HIP kernels.

Host-side mirror of nnU-Net `SegmentationNetwork.predict_3D` ->
`_internal_predict_3D_3Dconv_tiled` / `_internal_maybe_mirror_and_pred_3D`
[nnU-Net, not in tree; neural_network.py], with the options the reference's command lines fix:
mirroring on for `DeepWMH_predict` (/root/reference/deepwmh/main/predict.py:153-156), off with
`--disable_tta` (/root/reference/deepwmh/pipeline/DCNN_multistage.py:331-336), step 0.5, Gaussian.

Where nnU-Net runs 8 mirrored forwards one after another, the 8 mirrored copies form ONE batch
here (InstanceNorm is per sample, so the result is the same); without mirroring, up to
`batch` patches of the volume share a batch.  Patches are accumulated in nnU-Net's x->y->z order.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .network import LRELU_SLOPE, UNetEngine


def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """nnU-Net `_compute_steps_for_sliding_window` (Python round() is half-to-even like np.round)."""
    if any(i < p for i, p in zip(image_size, patch_size)):
        raise ValueError('image size must be as large or larger than patch_size')
    if not 0 < step_size <= 1:
        raise ValueError('step_size must be larger than 0 and smaller or equal to 1')
    steps = []
    for img, pat in zip(image_size, patch_size):
        target = pat * step_size
        num = int(math.ceil((img - pat) / target)) + 1
        max_step = img - pat
        actual = max_step / (num - 1) if num > 1 else 99999999999
        steps.append([int(round(actual * i)) for i in range(num)])
    return steps


def _gauss_kernel_1d(n, sigma):
    """Response along one axis of scipy.ndimage.gaussian_filter (truncate=4, mode='constant') to a
    delta at n//2: the normalised kernel of radius int(4*sigma+0.5), zero beyond it."""
    radius = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    out = np.zeros(n, dtype=np.float64)
    c = n // 2
    for i in range(n):
        d = i - c
        if -radius <= d <= radius:
            out[i] = phi[d + radius]
    return out


def get_gaussian(patch_size, sigma_scale=1. / 8):
    """nnU-Net `_get_gaussian`: the separable filter response, /max, float32, zeros -> min non-zero."""
    k = [_gauss_kernel_1d(int(p), p * sigma_scale) for p in patch_size]
    g = (k[0][:, None, None] * k[1][None, :, None]) * k[2][None, None, :]
    g = g / np.max(g) * 1
    g = g.astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


MIRROR_CODES = list(range(8))   # nnU-Net m = 0..7; bit0 = last axis, bit1 = middle, bit2 = first


class SlidingWindowPredictor:
    def __init__(self, engine: UNetEngine, patch_size=None, step_size=0.5, use_gaussian=True,
                 do_mirroring=True, batch=8):
        self.engine = engine
        self.lib = engine.lib
        self.device = engine.device
        self.patch = tuple(int(p) for p in (patch_size or engine.plan.patch_size))
        self.step_size = step_size
        self.use_gaussian = use_gaussian
        self.do_mirroring = do_mirroring
        self.batch = int(batch)
        assert 1 <= self.batch <= 64
        self._gauss = None
        self._cnt_cache = {}
        self._tmp = {}

    # ---- cached device constants -------------------------------------------------------------------
    def gaussian(self):
        if self._gauss is None:
            self._gauss = torch.from_numpy(get_gaussian(self.patch, 1. / 8)).to(self.device)
        return self._gauss

    def _geometry(self, vshape):
        A = tuple(max(v, p) for v, p in zip(vshape, self.patch))
        pad_lo = tuple((a - v) // 2 for a, v in zip(A, vshape))
        steps = compute_steps_for_sliding_window(self.patch, A, self.step_size)
        origins = [(x, y, z) for x in steps[0] for y in steps[1] for z in steps[2]]
        return A, pad_lo, origins

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _count_map(self, A, origins, gauss):
        """aggregated_nb_of_predictions: data independent, built once per geometry in patch order."""
        key = (A, tuple(origins), gauss is not None)
        cnt = self._cnt_cache.get(key)
        if cnt is None:
            cnt = torch.zeros(A, dtype=torch.float32, device=self.device)
            P = self.patch
            for o in origins:
                _lib.check(self.lib.wmh_accumulate_weight(_lib.ptr(gauss), _lib.ptr(cnt), P[0], P[1], P[2],
                                                          A[0], A[1], A[2], _lib.iarr(o), self._stream()),
                           'wmh_accumulate_weight')
            self._cnt_cache[key] = cnt
        return cnt

    def _scratch(self, key, shape, dtype=torch.float32):
        t = self._tmp.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._tmp[key] = t
        return t[:n].view(*shape)

    # ---- the hot path --------------------------------------------------------------------------------
    def predict_device(self, vol, return_prob=True):
        """vol: device tensor [C, X, Y, Z] fp32.  Returns (seg uint8 [X,Y,Z], prob [K,X,Y,Z] or None),
        both device tensors; everything is queued on the current stream (no host sync)."""
        assert vol.dim() == 4 and vol.dtype == torch.float32 and vol.is_contiguous()
        Cc = vol.shape[0]
        V = tuple(int(v) for v in vol.shape[1:])
        P = self.patch
        K = self.engine.plan.num_classes
        A, pad_lo, origins = self._geometry(V)
        gauss = self.gaussian() if (self.use_gaussian and len(origins) > 1) else None
        cnt = self._count_map(A, origins, gauss)
        agg = self._scratch('agg', (K,) + A)
        agg.zero_()
        st = self._stream()
        lib = self.lib

        def extract(n, orgs, flips):
            patches = self._scratch('patches', (n, Cc) + P)
            for k in range(0, n, 16):                     # a patch list holds WMH_MAX_BATCH_ITEMS = 16 items
                m = min(16, n - k)
                flat = [v for o in orgs[k:k + m] for v in o]
                _lib.check(lib.wmh_extract_patches(_lib.ptr(vol), _lib.ptr(patches[k:k + m]), Cc, V[0], V[1], V[2],
                                                   _lib.iarr(pad_lo), P[0], P[1], P[2], m, _lib.iarr(flat),
                                                   _lib.iarr(flips[k:k + m]), st), 'wmh_extract_patches')
            return patches

        def accumulate(y, ss, seg_w, origin, n, flips):
            if self.engine.acct is not None:          # the head reads the last feature map once (its output is in `agg`)
                self.engine.acct['flop'] += 2.0 * K * seg_w.shape[1] * n * P[0] * P[1] * P[2]
                self.engine.acct['bytes'] += y.element_size() * float(y[:n].numel())
            if self.engine.ncl:                       # CL16 bf16 feature map [n, C / 8, P0, P1, P2, 8]
                _lib.check(lib.wmh_cl_seghead_softmax_accumulate(
                    _lib.ptr(y), _lib.ptr(ss), _lib.ptr(seg_w), K, _lib.ptr(gauss), _lib.ptr(agg), 8 * y.shape[1],
                    P[0], P[1], P[2], A[0], A[1], A[2], _lib.iarr(origin), n, _lib.iarr(flips), LRELU_SLOPE, st),
                    'wmh_cl_seghead_softmax_accumulate')
                return
            _lib.check(lib.wmh_seghead_softmax_accumulate_st(
                _lib.ptr(y), _lib.ptr(ss), _lib.ptr(seg_w), K, _lib.ptr(gauss), _lib.ptr(agg), y.shape[1],
                P[0], P[1], P[2], A[0], A[1], A[2], _lib.iarr(origin), n, _lib.iarr(flips), LRELU_SLOPE,
                int(y.dtype == torch.bfloat16), st), 'wmh_seghead_softmax_accumulate_st')

        if self.do_mirroring:
            # the 8 mirrored copies of a tile are one batch of 8; `batch` >= 16 puts several tiles' copies into one forward (bigger grids
            # for the deep levels; the samples of a batch are independent, accumulation stays in tile order)
            tpf = max(1, self.batch // 8)
            for i in range(0, len(origins), tpf):
                grp = origins[i:i + tpf]
                n = 8 * len(grp)
                x = extract(n, [o for o in grp for _ in range(8)], list(MIRROR_CODES) * len(grp))
                y, ss, seg_w = self.engine.forward_features(x)[-1]
                for b, o in enumerate(grp):
                    accumulate(y[8 * b:8 * b + 8], ss[8 * b:8 * b + 8], seg_w, o, 8, MIRROR_CODES)
        else:
            for i in range(0, len(origins), self.batch):
                grp = origins[i:i + self.batch]
                n = len(grp)
                x = extract(n, grp, [0] * n)
                y, ss, seg_w = self.engine.forward_features(x)[-1]
                for b, o in enumerate(grp):       # fixed patch order: bit-reproducible accumulation
                    accumulate(y[b:b + 1], ss[b:b + 1], seg_w, o, 1, [0])

        seg = self._scratch('seg', V, torch.uint8)
        prob = self._scratch('prob', (K,) + V) if return_prob else None
        _lib.check(lib.wmh_finalize_argmax(_lib.ptr(agg), _lib.ptr(cnt), _lib.ptr(prob), _lib.ptr(seg), K,
                                           A[0], A[1], A[2], _lib.iarr(pad_lo), V[0], V[1], V[2], st),
                   'wmh_finalize_argmax')
        return seg, prob

    def warm_up(self, num_input_channels=None):
        """One forward on zeros at the batch size predict_device will use: allocates the engine's activation buffers (29 GB at 16 patches of
        the BASELINE plan) and makes the runtime load every kernel's code object -- half a second that `nnUNet_predict` otherwise spends inside its
        first case.  Queued on the current stream; no host sync."""
        Cc = int(num_input_channels or self.engine.plan.num_input_channels)
        n = 8 * max(1, self.batch // 8) if self.do_mirroring else min(self.batch, 8)
        x = self._scratch('patches', (n, Cc) + self.patch)
        x.zero_()
        self.engine.forward_features(x)
        self.gaussian()

    def predict(self, volume, return_prob=True):
        """volume: numpy float32 [C,X,Y,Z] (preprocessed).  Returns numpy (seg, prob)."""
        vol = torch.from_numpy(np.ascontiguousarray(volume, dtype=np.float32)).to(self.device)
        seg, prob = self.predict_device(vol, return_prob)
        torch.cuda.synchronize(self.device)
        return seg.cpu().numpy(), (prob.cpu().numpy() if prob is not None else None)
