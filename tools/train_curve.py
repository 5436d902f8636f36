"""Loss curves of the three training modes on the full-width network: the same 200 iterations -- BASELINE plan, batch 2 of
128^3 crops drawn from four synthetic FLAIR volumes with blob labels, poly learning rate from 1e-2, nnU-Net's He init -- in exact fp32, f16x2
(the default of nnUNet_train) and bf16 (opt-in), each from the same weights and the same patch sequence.  Writes the per-iteration losses and a
summary (mean of the last 20 iterations, largest deviation from the fp32 curve) as JSON:
    python tools/train_curve.py [iterations] > profiles/rNN_train_loss_curves.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd.plan import baseline_plan  # noqa: E402
from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair  # noqa: E402
from deepwmh_amd.trainer import UNetTrainer, downsample_targets, poly_lr  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
plan = baseline_plan()
sd = he_init_state_dict(plan, seed=0, trained_like=False)
vols = [synthetic_flair((160, 192, 160), seed=s)[0] for s in range(4)]
labs = [(v > 2.0).astype(np.uint8) for v in vols]            # the bright blobs of synthetic_flair as pseudo-lesions
rng = np.random.default_rng(0)
seq = [[(int(rng.integers(0, 4)), int(rng.integers(0, 33)), int(rng.integers(0, 65)), int(rng.integers(0, 33)), int(rng.integers(0, 8))) for _ in range(2)]
       for _ in range(iters)]


def batch(it):
    xs, ls = [], []
    for (v, x0, y0, z0, flip) in seq[it]:
        p, l = vols[v][x0:x0 + 128, y0:y0 + 128, z0:z0 + 128], labs[v][x0:x0 + 128, y0:y0 + 128, z0:z0 + 128]
        for ax in range(3):
            if (flip >> ax) & 1:
                p, l = np.flip(p, ax), np.flip(l, ax)
        xs.append(np.ascontiguousarray(p)); ls.append(np.ascontiguousarray(l))
    return torch.from_numpy(np.stack(xs)[:, None]).cuda(), torch.from_numpy(np.stack(ls)).cuda()


out = dict(workload='BASELINE plan (31.2 M parameters), batch 2 x 128^3 crops of 4 synthetic volumes (labels = their bright blobs), %d iterations, '
                    'poly LR from 1e-2, He init, identical patch sequence in every mode' % iters, modes={})
for prec in ('fp32', 'f16x2'):
    tr = UNetTrainer(plan, sd, 'cuda:0', precision=prec)
    losses = []
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(iters):
        x, lab = batch(it)
        td = downsample_targets(lab, plan.num_pool)
        losses.append(tr.run_iteration(x, td, poly_lr(it, iters, 1e-2, 0.9)).clone())
    torch.cuda.synchronize()
    dt = time.time() - t0
    ls = [float(v) for v in torch.stack(losses).cpu()]
    out['modes'][prec] = dict(losses=[round(v, 6) for v in ls], mean_last_20=float(np.mean(ls[-20:])), first=ls[0], skipped_steps=tr.skipped_steps,
                              wall_s_including_host_batching=round(dt, 1))
    del tr
    torch.cuda.empty_cache()
ref = np.array(out['modes']['fp32']['losses'])
for prec in ('f16x2',):
    cur = np.array(out['modes'][prec]['losses'])
    out['modes'][prec]['max_abs_diff_vs_fp32_curve'] = float(np.abs(cur - ref).max())
    out['modes'][prec]['mean_abs_diff_vs_fp32_curve_last_50'] = float(np.abs(cur - ref)[-50:].mean())
print(json.dumps(out))
