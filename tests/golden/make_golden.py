"""Generate the golden fixtures in this directory from the CPU oracle.

Run from the repo root:  python tests/golden/make_golden.py

The reference ships no tests/vectors for this path and its nnU-Net fork is absent,
so these vectors pin OUR oracle (torch-CPU restatement of nnU-Net v1), not reference outputs:
"parity unpinned" at the nnU-Net boundary.  Weights are stored (not re-drawn from a seed) so the
fixtures do not depend on torch's RNG stream.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepwmh_amd.plan import tiny_plan  # noqa: E402
from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy  # noqa: E402
from oracle import sliding_window as osw  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    net = perturb_like_trained(unet_from_plan(plan), seed=1).eval()
    sd = state_dict_numpy(net)
    np.savez_compressed(os.path.join(HERE, 'tiny_net_weights.npz'), **sd)

    vol = osw.synthetic_flair((40, 48, 40), seed=3)
    patch = np.ascontiguousarray(vol[:, 4:36, 8:40, 4:36])[None]
    with torch.no_grad():
        net.do_ds = True
        ds = net(torch.from_numpy(patch))
        net.do_ds = False
    out = dict(volume=vol, patch=patch)
    for i, t in enumerate(ds):
        out['logits_ds%d' % i] = t.numpy()
    seg_tta, prob_tta = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True)
    seg_nt, prob_nt = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, False, True)
    out.update(seg_tta=seg_tta, prob_tta=prob_tta, seg_notta=seg_nt, prob_notta=prob_nt)
    # a volume smaller than the patch in one axis (centred zero pad + crop) and a single-tile case
    small = np.ascontiguousarray(vol[:, :28, :, :32])
    seg_s, prob_s = osw.predict_3d_tiled(net, small, plan.patch_size, 0.5, False, True)
    out.update(small_volume=small, seg_small=seg_s, prob_small=prob_s)
    np.savez_compressed(os.path.join(HERE, 'tiny_net_io.npz'), **out)

    # steps table (G9) and Gaussian summaries (G10)
    table = []
    for img in [(160, 192, 160), (128, 128, 128), (129, 128, 255), (137, 201, 129), (40, 48, 40), (256, 256, 64)]:
        for pat in [(128, 128, 128), (32, 32, 32), (64, 96, 64)]:
            if all(i >= p for i, p in zip(img, pat)):
                for st in (0.5, 0.3, 1.0):
                    table.append(dict(image=img, patch=pat, step=st,
                                      steps=osw.compute_steps_for_sliding_window(pat, img, st)))
    import json
    with open(os.path.join(HERE, 'steps_table.json'), 'w') as f:
        json.dump(table, f)
    g32 = osw.get_gaussian((32, 32, 32))
    g128 = osw.get_gaussian((128, 128, 128))
    probes = [(0, 0, 0), (64, 64, 64), (63, 64, 65), (10, 100, 50), (127, 127, 127), (32, 96, 64), (1, 2, 3), (64, 0, 64)]
    np.savez_compressed(os.path.join(HERE, 'gaussian.npz'), g32=g32,
                        g128_probe_idx=np.array(probes), g128_probe_val=np.array([g128[p] for p in probes]),
                        g128_min=g128.min(), g128_max=g128.max(), g128_sum=np.float64(g128.astype(np.float64).sum()))
    print('wrote fixtures to', HERE)
    for f in sorted(os.listdir(HERE)):
        print('  %-28s %8d bytes' % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == '__main__':
    main()
