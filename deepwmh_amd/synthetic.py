"""Synthetic inputs for benchmarks and smoke runs (there is no dataset or checkpoint offline):
a FLAIR-like volume and random-init weights of the plan's architecture with
nnU-Net's InitWeights_He(1e-2) statistics, keyed like a `model_best.model` state dict."""
from collections import OrderedDict

import numpy as np


def synthetic_flair(shape=(160, 192, 160), seed=0):
    """Blurred noise inside an ellipsoid + a few bright blobs, z-scored over the non-zero voxels
    (nnU-Net's non-CT normalisation).  float32 [1, X, Y, Z]."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(1000 + seed)
    field = gaussian_filter(rng.standard_normal(shape), 4.0).astype(np.float32)
    field /= field.std()
    grids = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing='ij')
    c = [(s - 1) / 2 for s in shape]
    ax = [0.4375 * shape[0], 0.4427 * shape[1], 0.4375 * shape[2]]
    mask = sum(((g - ci) / ai) ** 2 for g, ci, ai in zip(grids, c, ax)) <= 1
    vol = field.copy()
    for _ in range(int(rng.integers(3, 11))):
        r = float(rng.uniform(2, 6))
        p = [rng.uniform(0.3 * s, 0.7 * s) for s in shape]
        vol[sum((g - pi) ** 2 for g, pi in zip(grids, p)) <= r * r] += 2.0
    vol = vol * mask
    nz = vol[mask]
    vol[mask] = (nz - nz.mean()) / (nz.std() + 1e-8)
    vol[~mask] = 0
    return vol[None].astype(np.float32)


def he_init_state_dict(plan, seed=0, trained_like=True):
    """Random weights: kaiming_normal(a=1e-2) conv / transposed-conv kernels, zero biases, IN affine
    (1, 0); `trained_like` perturbs biases and the IN affine so those paths carry signal."""
    rng = np.random.default_rng(seed)
    gain = np.sqrt(2.0 / (1 + 1e-2 ** 2))
    sd = OrderedDict()

    def conv_block(prefix, cin, cout, kz=3):
        sd[prefix + '.conv.weight'] = (rng.standard_normal((cout, cin, kz, 3, 3)) * gain / np.sqrt(cin * 9 * kz)).astype(np.float32)
        sd[prefix + '.conv.bias'] = (0.1 * rng.standard_normal(cout) if trained_like else np.zeros(cout)).astype(np.float32)
        sd[prefix + '.instnorm.weight'] = (1 + 0.2 * rng.standard_normal(cout) if trained_like else np.ones(cout)).astype(np.float32)
        sd[prefix + '.instnorm.bias'] = (0.1 * rng.standard_normal(cout) if trained_like else np.zeros(cout)).astype(np.float32)

    for s in plan.encoder_convs():
        conv_block(s.name, s.cin, s.cout, s.kz)
    dec = plan.decoder_stages()
    for st in dec:
        for c in st['convs']:
            conv_block(c.name, c.cin, c.cout, c.kz)
    for st in dec:
        name, cdown, cskip = st['tu']
        kt = int(st['tu_kt'])
        sd[name + '.weight'] = (rng.standard_normal((cdown, cskip, kt, 2, 2)) * gain / np.sqrt(cskip * 4 * kt)).astype(np.float32)
    for st in dec:
        cskip = st['tu'][2]
        sd[st['seg'] + '.weight'] = (rng.standard_normal((plan.num_classes, cskip, 1, 1, 1)) * gain / np.sqrt(cskip)).astype(np.float32)
    return sd
