"""Single-GPU rehearsal of the multi-GPU runs: a ONE-rank `nccl` (= RCCL) process group on the MI355X, and the
`force_collective` switch that makes PatchShardedPredictor / OverlappedGradientAverager issue their collectives although the world
size is 1.  The RCCL kernels (all_gather_into_tensor of the weighted tiles; asynchronous all_reduce of >= 8 MB runs of the gradient
arena) then execute beside the persistent one-workgroup-per-CU convolution kernels for the first time: stream ordering of
work.wait() against the accumulate kernels, the views handed to RCCL (alternating gather buffers, arena slices), and whether a
resident RCCL kernel and the persistent grids tolerate each other.  Results must equal the serial path bit for bit.

The process group is created before any other GPU call, in a fresh child process (no exec of a GPU-touched process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, socket
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)          # RCCL, before anything else touches the GPU
from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
from deepwmh_amd.network import UNetEngine
from deepwmh_amd.plan import baseline_plan, tiny_plan
from deepwmh_amd.sliding_window import SlidingWindowPredictor
from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
mode = sys.argv[1]
out = {}
if mode == 'predict':
    # full-width network (the persistent grids fill all CUs), a volume of 2 x 2 x 2 tiles: 8 rounds of forwards next to 8 all-gathers
    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2')
    vol = torch.from_numpy(synthetic_flair((160, 192, 160), seed=3)).cuda()
    for mirror in (False, True):
        pred = SlidingWindowPredictor(eng, do_mirroring=mirror, batch=8)
        seg0, prob0 = pred.predict_device(vol, return_prob=True)
        seg0, prob0 = seg0.clone(), prob0.clone()
        seg1, prob1 = PatchShardedPredictor(HipTileBackend(pred)).predict(vol)                           # serial shortcut (W == 1)
        seg1, prob1 = seg1.clone(), prob1.clone()
        seg2, prob2 = PatchShardedPredictor(HipTileBackend(pred), force_collective=True).predict(vol)    # through RCCL
        torch.cuda.synchronize()
        out['mirror%%d' %% mirror] = [bool(torch.equal(seg0, seg2) and torch.equal(prob0, prob2)),
                                      bool(torch.equal(seg1, seg2) and torch.equal(prob1, prob2)),
                                      float(prob2.sum()), bool(torch.isfinite(prob2).all()), float((prob0 - prob2).abs().max())]
elif mode == 'train':
    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).cuda()
    lab = (torch.rand(2, 128, 128, 128, generator=g) < 0.05).to(torch.uint8).cuda()
    td = downsample_targets(lab, plan.num_pool)
    res = []
    for force in (False, True):
        tr = UNetTrainer(plan, sd, 'cuda:0', precision='f16x2')
        for _ in range(2):                                       # two iterations: the second one's forward overlaps nothing stale
            tr.forward(x)
            loss = tr.backward(td, force_collective=force)
            tr.optimizer_step(1e-2)
        torch.cuda.synchronize()
        n_msgs = len(tr._avg.works) if hasattr(tr._avg, 'works') else -1
        res.append((float(loss.cpu()), tr.grads.clone(), tr.params.clone(), tr._avg.active))
        del tr
        torch.cuda.empty_cache()
    out['train'] = [bool(torch.equal(res[0][1], res[1][1])), bool(torch.equal(res[0][2], res[1][2])), res[0][0], res[1][0],
                    bool(res[0][3]), bool(res[1][3]), bool(torch.isfinite(res[1][1]).all())]
dist.barrier()
dist.destroy_process_group()
import json
print('RESULT ' + json.dumps(out))
'''


def _run(mode, timeout):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT), mode], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[len('RESULT '):])


@pytest.mark.timeout(900)
def test_patch_sharded_prediction_through_a_one_rank_rccl_group_is_bit_identical():
    out = _run('predict', 800)
    for key in ('mirror0', 'mirror1'):
        same_as_predictor, same_as_serial_shortcut, total, finite, diff = out[key]
        assert finite and total > 0
        assert same_as_serial_shortcut, '%s: the all-gather path differs from the serial shortcut of the same class' % key
        if key == 'mirror1':
            # with mirroring both paths run a tile's 8 mirrored copies as one forward: the same launches, the same bits
            assert same_as_predictor, '%s: the all-gather path differs from SlidingWindowPredictor' % key
        else:
            # without mirroring SlidingWindowPredictor batches the 8 TILES into one forward while the sharded path runs them one at a time:
            # other launch shapes (split-K of the deep levels), rounding-level differences (tests/test_gpu_network.py, property 4)
            assert diff <= 1e-5, diff


@pytest.mark.timeout(900)
def test_overlapped_gradient_all_reduce_through_a_one_rank_rccl_group_is_bit_identical():
    same_grads, same_params, loss0, loss1, active0, active1, finite = _run('train', 800)['train']
    assert not active0 and active1, 'force_collective must be what switches the all-reduces on at world size 1'
    assert finite and loss0 == loss1
    assert same_grads and same_params, 'gradients / parameters after two iterations differ between the forced RCCL path and the plain one'
