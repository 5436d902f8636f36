"""configs[1] exactly as `DeepWMH_predict` runs it (/root/reference/deepwmh/main/predict.py:153-156: mirroring ON) at full size,
computed ONCE by the CPU oracle in the build container: 160x192x160 volume, 8 tiles of 128^3, 8 mirrored forwards each
(64 forwards of the 31 M-parameter net; about an hour on 8 cores), Gaussian aggregation, argmax.

Output tests/golden/fullsize_tta.npz (compact, so that the -m gpu suite can check the whole mirrored volume without re-running the
oracle for an hour on the GPU box's host):
  mask_bits      np.packbits of the oracle's uint8 label map
  p1_lattice     softmax channel 1 on the 1-in-64 lattice [2::4, 2::4, 2::4] (float32)
  band_index     flat indices of the voxels whose oracle class margin |p1 - p0| is <= 1e-3, band_p1 their probabilities
                 (everything the argmax comparison needs to exclude a margin band of any width up to 1e-3)
  sum_p1, sum_p1_sq   float64 checksums of channel 1 over the whole volume
  input checksums (volume, state dict) so the test knows it rebuilt the same inputs.

Run from the repo root:  python tests/golden/make_golden_fullsize.py [--net perturbed|he]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def build_net(kind):
    """The two BASELINE-architecture networks of tests/test_gpu_network.py: 'perturbed' = He init + trained-like perturbation
    (fixture `baseline_pair`), 'he' = plain InitWeights_He(1e-2) (what bench.py's synthetic weights look like)."""
    from deepwmh_amd.plan import baseline_plan
    from oracle.generic_unet import perturb_like_trained, unet_from_plan
    torch.manual_seed(0)
    plan = baseline_plan()
    net = unet_from_plan(plan)
    if kind == 'perturbed':
        net = perturb_like_trained(net, seed=2)
    net = net.eval()
    net.do_ds = False
    return plan, net


def state_checksum(net):
    return float(sum(float(v.double().abs().sum()) for v in net.state_dict().values()))


def main():
    from oracle import sliding_window as osw
    kind = 'perturbed'
    if '--net' in sys.argv:
        kind = sys.argv[sys.argv.index('--net') + 1]
    plan, net = build_net(kind)
    vol = osw.synthetic_flair((160, 192, 160), seed=3)
    t0 = time.time()
    done = []

    def hook(origin, pred):
        done.append(origin)
        print('tile %d/8 at %s after %.0f s' % (len(done), origin, time.time() - t0), flush=True)

    seg, prob = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True, patch_hook=hook)
    margin = np.abs(prob[1] - prob[0]).reshape(-1)
    band = np.flatnonzero(margin <= 1e-3).astype(np.int32)
    out = dict(mask_bits=np.packbits(seg.astype(np.uint8).reshape(-1)), shape=np.array(seg.shape),
               p1_lattice=prob[1][2::4, 2::4, 2::4].astype(np.float32), band_index=band, band_p1=prob[1].reshape(-1)[band].astype(np.float32),
               sum_p1=np.float64(prob[1].astype(np.float64).sum()), sum_p1_sq=np.float64((prob[1].astype(np.float64) ** 2).sum()),
               foreground=np.int64(seg.sum()), volume_abs_sum=np.float64(np.abs(vol.astype(np.float64)).sum()),
               state_abs_sum=np.float64(state_checksum(net)), oracle_seconds=np.float64(time.time() - t0),
               torch_version=np.array(torch.__version__))
    name = 'fullsize_tta.npz' if kind == 'perturbed' else 'fullsize_tta_%s.npz' % kind
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('wrote %s: %.0f KB, %d foreground voxels, %d voxels inside the 1e-3 margin band, %.0f s of oracle time'
          % (name, os.path.getsize(os.path.join(HERE, name)) / 1e3, int(seg.sum()), len(band), time.time() - t0))


if __name__ == '__main__':
    main()
