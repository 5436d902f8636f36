"""Whole-volume passes at the BASELINE volume size: device time per call (HIP events on the
launch stream, inputs resident in HBM), achieved algorithmic GB/s against the 8 TB/s HBM roofline, and the CPU oracle
beside it.  usage: python tools/volume_ops_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepwmh_amd import _lib, postprocess, stage2          # noqa: E402
from deepwmh_amd.synthetic import synthetic_flair            # noqa: E402
from oracle import postprocess as opp, stage2 as o2          # noqa: E402

lib = _lib.load()
P = _lib.ptr
HBM = 8000.0


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cpu_time(fn, reps=3):
    fn()
    t = time.time()
    for _ in range(reps):
        fn()
    return (time.time() - t) / reps * 1e3


def row(name, ms, bytes_alg, cpu_ms):
    gbs = bytes_alg / ms / 1e6
    print('%-34s %8.3f ms  %8.1f GB/s  (%.3f of HBM)   cpu oracle %9.1f ms   x%.0f' % (name, ms, gbs, gbs / HBM, cpu_ms, cpu_ms / ms))


shape = (160, 192, 160)
n = int(np.prod(shape))
v = synthetic_flair(shape, 0)[0]
raw = (v * 50 + 300).astype(np.float32); raw[v == 0] = 0
lesion = (v > 1.2).astype(np.uint8)
st = _lib.current_stream()

md = torch.from_numpy(lesion).cuda(); out = torch.empty_like(md)
lab = torch.empty(n, dtype=torch.int32, device='cuda'); cnt = torch.empty_like(lab)
ms = timed(lambda: _lib.check(lib.wmh_remove_small_components(P(md), P(out), P(lab), P(cnt), *shape, 3, st)))
# algorithmic bytes: read mask, write out (1 B each) + one int32 label image written and read once
row('remove_3mm_sparks (lesion mask)', ms, n * (1 + 1 + 4 + 4), cpu_time(lambda: opp.remove_sparks(lesion, 3)))

brain = torch.from_numpy((raw != 0).astype(np.uint8)).cuda()
ms = timed(lambda: _lib.check(lib.wmh_fill_holes(P(brain), P(out), P(lab), P(cnt), *shape, st)))
row('fill_holes (brain mask)', ms, n * (1 + 1 + 4 + 4), cpu_time(lambda: opp.fill_holes(brain.cpu().numpy())))

x = torch.from_numpy(raw[None]).cuda(); bbox = torch.empty(6, dtype=torch.int32, device='cuda')
_lib.check(lib.wmh_mask_bbox(P(brain), P(bbox), *shape, st))
bb = bbox.cpu().tolist(); cshape = (bb[1] - bb[0], bb[3] - bb[2], bb[5] - bb[4]); nc = int(np.prod(cshape))
o = torch.empty((1,) + cshape, device='cuda'); stats = torch.empty(2, device='cuda'); ws = torch.empty(4096, dtype=torch.float64, device='cuda')
ms = timed(lambda: _lib.check(lib.wmh_zscore_crop(P(x), P(brain), P(bbox), P(o), None, P(stats), P(ws), 1, *shape, 1, st)))
from deepwmh_amd import preprocess
bm = brain.cpu().numpy().astype(bool)
sl = tuple(slice(bb[2 * a], bb[2 * a + 1]) for a in range(3))
row('z-score + crop', ms, nc * (4 + 1) * 2 + nc * 4, cpu_time(lambda: preprocess.normalize_noncT(raw[None][(slice(None),) + sl], bm[sl], True)))

E = 5
rng = np.random.default_rng(0)
sm = [rng.random(shape).astype(np.float32) for _ in range(E)]
stack = torch.stack([torch.from_numpy(s) for s in sm]).cuda(); field = torch.empty(shape, device='cuda'); lab8 = torch.empty(shape, dtype=torch.uint8, device='cuda')
ms = timed(lambda: _lib.check(lib.wmh_ensemble_softmax(P(stack), E, P(field), P(lab8), n, st)))
def cpu_ens():
    f = np.zeros(shape).astype('float32')
    for s in sm:
        f += s
    f = f / E
    return (f < 0.5).astype('float32')
row('ensemble 5 softmaxes + label', ms, n * (4 * E + 4 + 1), cpu_time(cpu_ens))
m = torch.from_numpy((rng.random(shape) < 0.8).astype(np.float32)).cuda(); y = torch.empty_like(m)
ms = timed(lambda: _lib.check(lib.wmh_softmax_mask(P(stack[0]), P(m), P(y), n, st)))
mm = m.cpu().numpy()
row('softmax masking', ms, n * 12, cpu_time(lambda: o2.softmax_masking(sm[0], mm)))
o3 = torch.zeros(3, dtype=torch.int64, device='cuda')
ms = timed(lambda: _lib.check(lib.wmh_binary_overlap(P(stack[0]), P(stack[1]), P(o3), n, st)))
row('hard Dice counts', ms, n * 8, cpu_time(lambda: o2.hard_dice_binary(sm[0], sm[1])))

from deepwmh_amd import preprocess as pre
new = (144, 173, 192)
xr = torch.from_numpy(raw[None]).cuda()
ws = torch.empty(lib.wmh_resample_workspace_doubles(*shape, 3), dtype=torch.float64, device='cuda'); outr = torch.empty((1,) + new, device='cuda')
ms = timed(lambda: _lib.check(lib.wmh_resample(P(xr), P(outr), P(ws), 1, *shape, *new, 3, st)), reps=5)
nn_ = int(np.prod(new)); npad = int(np.prod([v + 24 for v in shape]))
# algorithmic bytes: read the volume, write + 3x (read, write) the double coefficients, read them once more, write the output
row('resample order 3 (spline)', ms, n * 4 + npad * 8 * 8 + nn_ * 4, cpu_time(lambda: pre._resize(raw.astype(np.float64), new, 3), reps=1))
pr = torch.rand((2,) + new, device='cuda'); outp = torch.empty((2,) + shape, device='cuda')
ms = timed(lambda: _lib.check(lib.wmh_resample(P(pr), P(outp), None, 2, *new, *shape, 1, st)), reps=5)
prh = pr.cpu().numpy()
row('resample order 1 (2 softmax ch.)', ms, 2 * (nn_ + n) * 4, cpu_time(lambda: [pre._resize(prh[k].astype(np.float64), shape, 1) for k in range(2)], reps=1))
