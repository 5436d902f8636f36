"""Synthetic Stage-I case shared by the CPU and GPU tests: a target with a bright lesion, R registered references,
brain masks and tissue labels (0 background, 1 cerebrum, 2 cerebellum/brainstem, 3 cortex)."""
import numpy as np
from scipy.ndimage import gaussian_filter


def make_case(shape=(40, 44, 36), R=6, seed=0):
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing='ij')
    brain = sum(((g - (s - 1) / 2) / (0.42 * s)) ** 2 for g, s in zip(grids, shape)) <= 1

    def img(sd, lesion=False):
        r = np.random.default_rng(sd)
        v = gaussian_filter(r.standard_normal(shape), 2.0) * 40 + 300 + 30 * np.sin(grids[0] / 6.0)
        if lesion:
            v[18:22, 20:25, 15:19] += 200
        return (v * brain).astype(np.float32)

    x = img(seed + 1, True)
    refs = [img(seed + 10 + i) for i in range(R)]
    rng = np.random.default_rng(seed + 99)
    l1 = []
    for _ in range(R):
        m = brain.copy()
        m ^= (rng.random(shape) < 0.01) & brain                   # the references' masks disagree a little
        l1.append(m.astype(np.float32))
    tissue = np.where(brain, 1 + (grids[0] > 28) * 1, 0)
    tissue = np.where(brain & (grids[1] < 10), 3, tissue).astype(np.float32)
    l2 = []
    for _ in range(R):
        t = tissue.copy()
        flip = rng.random(shape) < 0.02
        t[flip] = rng.integers(0, 4, int(flip.sum()))
        l2.append(t.astype(np.float32))
    return x, refs, l1, l2
