"""W > 1 on the HIP kernels where no multi-GPU node exists: two and three ranks that SHARE the one MI355X, gloo between
them (RCCL wants one GPU per rank; device tensors are staged through the host for the all-gather, distributed.py).  The collectives are not
RCCL's -- that path is rehearsed with a one-rank group in tests/test_gpu_rccl.py -- but everything else is the multi-GPU run: every rank
computes only its share of the patches on the HIP kernels, the tiles meet in the exchange, every rank accumulates all of them in nnU-Net's
order; the training ranks see different batches and average their gradient arenas under the backward pass.
  * patch-sharded prediction at W = 2 and W = 3 (ragged last round) is bit-identical to the serial predictor on every rank;
  * data-parallel training: the averaged arena is the mean of the ranks' local arenas (bit for bit: an element-wise sum of two floats, halved), the
    parameters stay identical on all ranks over the steps, and differ from a rank's single-process trajectory (the exchange really happened)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world, mode = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), sys.argv[1]
torch.cuda.set_device(0)                                   # every rank on the same GPU
dist.init_process_group('gloo', rank=rank, world_size=world)
from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
from deepwmh_amd.network import UNetEngine
from deepwmh_amd.plan import tiny_plan
from deepwmh_amd.sliding_window import SlidingWindowPredictor
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
golden = os.path.join(%(root)r, 'tests', 'golden')
plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
sd = dict(np.load(os.path.join(golden, 'tiny_net_weights.npz')))
io = np.load(os.path.join(golden, 'tiny_net_io.npz'))
out = {}
if mode == 'predict':
    eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2')
    vol = torch.from_numpy(io['volume']).cuda()            # 40 x 48 x 40: 2 x 2 x 2 tiles of 32^3
    for mirror in (False, True):
        pred = SlidingWindowPredictor(eng, do_mirroring=mirror, batch=8)
        sharded = PatchShardedPredictor(HipTileBackend(pred))
        assert sharded.world == world and sharded.rank == rank
        seg1, prob1 = sharded.predict(vol)
        seg1, prob1 = seg1.clone(), prob1.clone()
        # the serial loop of the same class (one rank, no group): tile by tile, the launches every rank's share consists of
        serial = PatchShardedPredictor.__new__(PatchShardedPredictor)
        serial.b, serial.group, serial.force, serial.rank, serial.world, serial.stall_marks = HipTileBackend(pred), None, False, 0, 1, None
        seg0, prob0 = serial.predict(vol)
        torch.cuda.synchronize()
        ref = prob1.clone().cpu()
        dist.broadcast(ref, 0)                              # and all ranks hold the same result
        out['mirror%%d' %% mirror] = [bool(torch.equal(seg0, seg1) and torch.equal(prob0, prob1)), bool(torch.equal(ref, prob1.cpu())),
                                      float(prob1.sum()), float((prob1 - torch.from_numpy(io['prob_tta' if mirror else 'prob_notta']).cuda()).abs().max())]
elif mode == 'train':
    vol = io['volume'][0]
    crops = [vol[4:36, 8:40, 4:36], vol[8:40, 0:32, 2:34][::-1].copy(), vol[2:34, 16:48, 6:38], vol[0:32, 4:36, 8:40][:, ::-1].copy()]
    data = np.stack(crops[2 * rank:2 * rank + 2])[:, None].astype(np.float32)          # every rank its own batch of 2
    labels = (data[:, 0] > (0.3 + 0.2 * rank)).astype(np.uint8)
    x = torch.from_numpy(data).cuda()
    td = downsample_targets(torch.from_numpy(labels).cuda(), plan.num_pool)
    groups = [dist.new_group([r]) for r in range(world)]    # (every rank must take part in every new_group call)
    tr = UNetTrainer(plan, sd, 'cuda:0', precision='f16x2')
    # local gradients: the same backward pass against a ONE-rank subgroup (no exchange)
    tr.forward(x); tr.backward(td, groups[rank]); local = tr.grads.clone()
    tr.forward(x); loss = tr.backward(td, None); avg = tr.grads.clone()               # the world group: averaged on the way
    torch.cuda.synchronize()
    gathered = [torch.empty_like(local.cpu()) for _ in range(world)]
    dist.all_gather(gathered, local.cpu())
    mean = gathered[0].clone()
    for g in gathered[1:]:
        mean += g
    mean /= world
    same_mean = bool(torch.equal(mean, avg.cpu()))
    differs_from_local = bool(not torch.equal(local, avg))
    for _ in range(3):
        tr.run_iteration(x, td, 1e-2)
    torch.cuda.synchronize()
    p = tr.params.cpu()
    ref = p.clone(); dist.broadcast(ref, 0)
    trs = UNetTrainer(plan, sd, 'cuda:0', precision='f16x2')                         # a rank on its own: another trajectory
    for _ in range(3):
        trs.run_iteration(x, td, 1e-2, group=groups[rank])
    torch.cuda.synchronize()
    out['train'] = [same_mean, differs_from_local, bool(torch.equal(ref, p)), bool(not torch.equal(trs.params.cpu(), p)), float(loss.cpu()),
                    bool(torch.isfinite(p).all())]
dist.barrier()
dist.destroy_process_group()
print('RESULT ' + json.dumps(out), flush=True)
'''


def _run(mode, world, timeout=600):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-c', CHILD % dict(root=ROOT), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, (so[-1500:], se[-3000:])
        outs.append(json.loads([l for l in so.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):]))
    return outs


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [2, 3])
def test_patch_sharded_prediction_of_w_ranks_on_the_hip_kernels_is_bit_identical(world):
    for out in _run('predict', world):
        for key in ('mirror0', 'mirror1'):
            same_as_serial, same_on_all_ranks, total, err_vs_oracle = out[key]
            assert same_as_serial, '%s: W = %d differs from the serial loop' % (key, world)
            assert same_on_all_ranks and total > 0
            assert err_vs_oracle <= 1e-4                    # and it is the oracle's prediction (golden fixture)


@pytest.mark.timeout(900)
def test_data_parallel_training_of_two_ranks_on_the_hip_kernels():
    for out in _run('train', 2):
        same_mean, differs_from_local, same_params, differs_from_solo, loss, finite = out['train']
        assert finite and loss == loss
        assert same_mean, 'the averaged arena is not the mean of the two local arenas'
        assert differs_from_local and differs_from_solo, 'no exchange took place'
        assert same_params, 'the ranks drifted apart'


@pytest.mark.timeout(900)
def test_nnunet_train_data_parallel_command_with_two_ranks_sharing_the_gpu(tmp_path):
    """configs[4] behind the reference's boundary: `torch.distributed.run --nproc-per-node 2 nnUNet_train ...` (the reference itself passes one GPU:
    /root/reference/deepwmh/pipeline/DCNN_multistage.py:292-309), both ranks on the one MI355X (WMH_TRAIN_BACKEND=gloo).  The command must train,
    average, write rank 0's checkpoints and exit 0; the checkpoint must load and a single-rank `-c` run must continue from it."""
    import shlex
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_train_pipeline import make_raw_dataset
    raw, pre, res = str(tmp_path / 'raw'), str(tmp_path / 'pre'), str(tmp_path / 'results')
    task = make_raw_dataset(raw, n=3)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, nnUNet_raw_data_base=raw, nnUNet_preprocessed=pre, RESULTS_FOLDER=res, CUDA_VISIBLE_DEVICES='0',
               PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ['PATH'], WMH_TRAIN_BACKEND='gloo', OMP_NUM_THREADS='4')
    p = subprocess.run(shlex.split('nnUNet_plan_and_preprocess -t 1 --verify_dataset_integrity'), env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-1500:]
    train = 'nnUNet_train 3d_fullres nnUNetTrainerV2 1 all -e 2 -b 3 --noval --save_every_epoch --disable_postprocessing_on_folds --val_batches 2'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bin', 'nnUNet_train')] + shlex.split(train)[1:]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert '2 rank(s)' in p.stdout and 'epoch 2/2' in p.stdout and p.stdout.count('epoch 2/2') == 1          # rank 0 reports, once
    fdir = os.path.join(res, 'nnUNet', '3d_fullres', task, 'nnUNetTrainerV2__nnUNetPlansv2.1', 'all')
    for f in ('model_latest.model', 'model_best.model', 'model_ep_0002.model', 'model_final_checkpoint.model'):
        assert os.path.isfile(os.path.join(fdir, f)), f
    from deepwmh_amd import checkpoint as ck
    _, _, sd, info = ck.load_checkpoint(os.path.dirname(fdir), 'all', 'model_latest')
    assert info['epoch'] == 2 and all(np.isfinite(v).all() for v in sd.values())
    env1 = {k: v for k, v in env.items() if k != 'WMH_TRAIN_BACKEND'}
    q = subprocess.run(shlex.split(train.replace('-e 2', '-e 3') + ' -c'), env=env1, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0 and 'continuing from' in q.stdout and 'epoch 3/3' in q.stdout, (q.stdout[-800:], q.stderr[-1500:])
