"""The "trained-like" fixture SURVEY.md 8c/8d lists: the committed tiny network after 20 SGD steps of the ORACLE's training step.

Run from the repo root (build container; CPU, ~20 s):  python tests/golden/make_golden_trained_tiny.py

Starting from tests/golden/tiny_net_weights.npz, `oracle/training.py::run_iteration` (our torch-autograd restatement of
nnUNetTrainerV2.run_iteration: deep-supervision Dice+CE, clip 12, SGD-Nesterov 0.99, weight decay 3e-5, poly learning rate from 1e-2
over 20 steps) is run 20 times on batches of two 32^3 patches cut deterministically from the committed 40x48x40 volume
(tests/golden/tiny_net_io.npz); labels = "volume > 0.5" (about a third of the brain voxels: a mask the net can learn and whose argmax
is not degenerate).  Stored: the patch schedule, the 20 loss values, the parameters after step 20, and the trained network's tiled
prediction of the whole volume (with and without mirroring).  Like every network-path fixture this pins OUR oracle -- the nnU-Net
fork is absent from /root/reference ("parity unpinned").

Consumers: tests/test_oracle.py (the fixture is what the oracle computes today), tests/test_gpu_trained.py (the HIP trainer follows the
oracle's 20 steps; the HIP predictor reproduces the trained net's probabilities and label map).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepwmh_amd.plan import tiny_plan  # noqa: E402
from oracle import sliding_window as osw  # noqa: E402
from oracle import training as ot  # noqa: E402
from oracle.generic_unet import state_dict_numpy, unet_from_plan  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS = 20
THRESHOLD = 0.5


def schedule(steps=STEPS, seed=11):
    """[steps][2] patches: (x0, y0, z0, flip bits) into the 40x48x40 volume."""
    rng = np.random.default_rng(seed)
    return np.array([[(int(rng.integers(0, 9)), int(rng.integers(0, 17)), int(rng.integers(0, 9)), int(rng.integers(0, 8))) for _ in range(2)]
                     for _ in range(steps)], dtype=np.int32)


def batch(vol, lab, sched_row):
    xs, ls = [], []
    for (x0, y0, z0, flip) in sched_row:
        p, l = vol[x0:x0 + 32, y0:y0 + 32, z0:z0 + 32], lab[x0:x0 + 32, y0:y0 + 32, z0:z0 + 32]
        for ax in range(3):
            if (int(flip) >> ax) & 1:
                p, l = np.flip(p, ax), np.flip(l, ax)
        xs.append(np.ascontiguousarray(p)); ls.append(np.ascontiguousarray(l))
    return np.stack(xs)[:, None].astype(np.float32), np.stack(ls).astype(np.uint8)


def train_oracle(plan, sd0, vol, lab, sched):
    net = unet_from_plan(plan)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()}, strict=True)
    net.train()
    weights = ot.deep_supervision_weights(plan.num_pool)
    opt = ot.make_optimizer(net, 1e-2)
    losses = []
    for it in range(len(sched)):
        for g in opt.param_groups:
            g['lr'] = ot.poly_lr(it, len(sched), 1e-2, 0.9)
        x, l = batch(vol, lab, sched[it])
        tgt = ot.downsample_targets(torch.from_numpy(l[:, None].astype(np.float32)), plan.num_pool)
        losses.append(ot.run_iteration(net, opt, torch.from_numpy(x), tgt, weights))
    return net.eval(), np.array(losses, dtype=np.float64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    sd0 = dict(np.load(os.path.join(HERE, 'tiny_net_weights.npz')))
    vol4 = np.load(os.path.join(HERE, 'tiny_net_io.npz'))['volume']
    vol = vol4[0]
    lab = (vol > THRESHOLD).astype(np.uint8)
    sched = schedule()
    net, losses = train_oracle(plan, sd0, vol, lab, sched)
    out = {'w.' + k: v for k, v in state_dict_numpy(net).items()}
    net.do_ds = False
    seg_tta, prob_tta = osw.predict_3d_tiled(net, vol4, plan.patch_size, 0.5, True, True)
    seg_nt, prob_nt = osw.predict_3d_tiled(net, vol4, plan.patch_size, 0.5, False, True)
    out.update(schedule=sched, losses=losses, threshold=np.float32(THRESHOLD), label_voxels=np.int64(lab.sum()),
               seg_tta=seg_tta, prob1_tta=prob_tta[1], seg_notta=seg_nt, prob1_notta=prob_nt[1])
    np.savez_compressed(os.path.join(HERE, 'tiny_net_trained20.npz'), **out)
    print('losses:', np.round(losses, 4))
    print('labels %d voxels of %d; trained net predicts %d (mirroring) / %d; Dice vs labels %.4f; |p1 - 0.5| < 1e-3 on %d voxels'
          % (lab.sum(), lab.size, seg_tta.sum(), seg_nt.sum(), osw.hard_dice_binary(lab, seg_tta), int((np.abs(prob_tta[1] - 0.5) < 1e-3).sum())))
    print('%d bytes' % os.path.getsize(os.path.join(HERE, 'tiny_net_trained20.npz')))


if __name__ == '__main__':
    main()
