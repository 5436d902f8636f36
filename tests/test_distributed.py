"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: case sharding and the patch-sharded
all-gather mode, with the oracle standing in for the HIP arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepwmh_amd.distributed import PatchShardedPredictor, shard_cases


def test_shard_cases_partition():
    cases = ['c%d' % i for i in range(7)]
    parts = [shard_cases(cases, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(cases) and parts[0] == ['c0', 'c3', 'c6']
    assert shard_cases([], 0, 2) == []


class OracleTileBackend:
    """CPU stand-in: same interface as HipTileBackend, arithmetic by the oracle."""

    def __init__(self, net, patch, mirror):
        self.net, self.patch, self.mirror = net, tuple(patch), mirror

    def geometry(self, V):
        from deepwmh_amd.sliding_window import compute_steps_for_sliding_window, get_gaussian
        A = tuple(max(v, p) for v, p in zip(V, self.patch))
        self.pad_lo = tuple((a - v) // 2 for a, v in zip(A, V))
        st = compute_steps_for_sliding_window(self.patch, A, 0.5)
        origins = [(x, y, z) for x in st[0] for y in st[1] for z in st[2]]
        self.gauss = torch.from_numpy(get_gaussian(self.patch)) if len(origins) > 1 else None
        return A, self.pad_lo, origins, self.net.num_classes, self.patch

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float32)

    def tile(self, vol, origin):
        from oracle.sliding_window import mirror_and_predict, pad_nd_image
        data, _ = pad_nd_image(vol.numpy(), self.patch)
        x, y, z = origin
        P = self.patch
        patch = torch.from_numpy(np.ascontiguousarray(data[None, :, x:x + P[0], y:y + P[1], z:z + P[2]]))
        return mirror_and_predict(self.net, patch, self.mirror, self.gauss)[0]

    def accumulate(self, agg, tile, origin):
        x, y, z = origin
        P = self.patch
        agg[:, x:x + P[0], y:y + P[1], z:z + P[2]] += tile

    def count_map(self, A, origins):
        cnt = torch.zeros(A)
        P = self.patch
        for (x, y, z) in origins:
            cnt[x:x + P[0], y:y + P[1], z:z + P[2]] += self.gauss if self.gauss is not None else 1.0
        return cnt

    def finalize(self, agg, cnt, pad_lo, V):
        sl = tuple(slice(l, l + v) for l, v in zip(pad_lo, V))
        prob = (agg / cnt)[(slice(None),) + sl]
        return prob.argmax(0).to(torch.uint8), prob


def _worker(rank, world, port, golden, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deepwmh_amd.plan import tiny_plan
    from oracle.generic_unet import unet_from_plan
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    sd = dict(np.load(os.path.join(golden, 'tiny_net_weights.npz')))
    net = unet_from_plan(plan).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    io = np.load(os.path.join(golden, 'tiny_net_io.npz'))
    pred = PatchShardedPredictor(OracleTileBackend(net, plan.patch_size, False))
    seg, prob = pred.predict(torch.from_numpy(io['volume']))
    # every rank holds the full result; they must agree bit for bit
    ref = prob.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, prob)
    if rank == 0:
        np.save(out, prob.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_patch_sharded_predict_gloo(world, golden_dir, tmp_path, tiny_io):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'prob.npy')
    mp.spawn(_worker, args=(world, port, golden_dir, out), nprocs=world, join=True)
    prob = np.load(out)
    # 8 patches over 2 (even) or 3 (ragged last round) ranks == the serial oracle, bit for bit
    assert np.array_equal(prob, tiny_io['prob_notta']) or float(np.abs(prob - tiny_io['prob_notta']).max()) <= 2e-6


def _grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepwmh_amd.distributed import average_gradients
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    a = average_gradients(g.clone(), None)
    b = average_gradients(g.clone(), None, bucket_elems=96)            # ragged buckets, issued back to front
    assert torch.equal(a, b)
    if rank == 0:
        np.save(out, a.numpy())
    dist.destroy_process_group()


def test_gradient_averaging_gloo(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'g.npy')
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), np.arange(1000, dtype=np.float32) * 1.5)


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepwmh_amd.distributed import OverlappedGradientAverager, average_gradients
    # an arena of 7 parameters with alignment padding inside the ranges, reported in backward order (not arena order)
    sizes = [40, 8, 400, 12, 1000, 4, 36]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ranges = {'p%d' % i: (int(offs[i]), int(offs[i + 1])) for i in range(len(sizes))}
    g = torch.arange(int(offs[-1]), dtype=torch.float32) * (rank + 1) + rank
    want = average_gradients(g.clone(), None)
    ov = OverlappedGradientAverager(g, ranges, None, min_elems=300)
    ov.ready(['p6', 'p5'])            # small: stays pending
    ov.ready(['p4'])                  # merges with p5, p6 into one run >= 300 -> sent
    ov.ready(['p1'])
    ov.ready(['p2', 'p3'])            # p1..p3 merge -> sent
    ov.ready(['p0'])
    got = ov.finish()
    assert torch.equal(got, want)
    assert len(ov.works) == 0 and ov.sent == 0          # ready for the next iteration
    # protocol errors are loud
    ov.ready(['p0'])
    try:
        ov.ready(['p0'])
        raise AssertionError('double report accepted')
    except RuntimeError:
        pass
    ov.begin()
    ov.ready(['p0', 'p1'])
    try:
        ov.finish()
        raise AssertionError('incomplete backward accepted')
    except RuntimeError:
        pass
    if rank == 0:
        np.save(out, got.numpy())
    dist.destroy_process_group()


def test_overlapped_gradient_averaging_gloo(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'g.npy')
    mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
    n = 1500
    assert np.array_equal(np.load(out), (np.arange(n, dtype=np.float32) * 3 + 1) / 2)


def test_overlapped_gradient_averaging_single_process_is_a_checked_no_op():
    from deepwmh_amd.distributed import OverlappedGradientAverager
    g = torch.arange(10, dtype=torch.float32)
    ov = OverlappedGradientAverager(g, {'a': (0, 4), 'b': (4, 10)})
    ov.ready(['b']); ov.ready(['a'])
    assert torch.equal(ov.finish(), torch.arange(10, dtype=torch.float32))
    ov.ready(['a'])
    with pytest.raises(RuntimeError):
        ov.finish()


def _forced_worker(rank, world, port, golden, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deepwmh_amd.distributed import OverlappedGradientAverager
    from deepwmh_amd.plan import tiny_plan
    from oracle.generic_unet import unet_from_plan
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    sd = dict(np.load(os.path.join(golden, 'tiny_net_weights.npz')))
    net = unet_from_plan(plan).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    io = np.load(os.path.join(golden, 'tiny_net_io.npz'))
    vol = torch.from_numpy(io['volume'])
    plain = PatchShardedPredictor(OracleTileBackend(net, plan.patch_size, False))
    forced = PatchShardedPredictor(OracleTileBackend(net, plan.patch_size, False), force_collective=True)
    assert not plain.force and forced.force
    (s0, p0), (s1, p1) = plain.predict(vol), forced.predict(vol)
    assert torch.equal(p0, p1) and torch.equal(s0, s1)
    g = torch.arange(1500, dtype=torch.float32) + 1
    ov = OverlappedGradientAverager(g.clone(), {'a': (0, 400), 'b': (400, 1500)}, None, min_elems=300, force_collective=True)
    assert ov.active and ov.world == 1
    ov.ready(['b']); ov.ready(['a'])
    assert torch.equal(ov.finish(), g)                       # a sum over one rank, not divided
    np.save(out, p1.numpy())
    dist.destroy_process_group()


def test_force_collective_at_world_size_one_gloo(golden_dir, tmp_path, tiny_io):
    """the switch that makes the one-rank case go through all_gather_into_tensor / async all_reduce (the GPU test
    tests/test_gpu_rccl.py runs it on a one-rank RCCL group); here on gloo: same bits as the serial shortcut and as the oracle."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'prob.npy')
    mp.spawn(_forced_worker, args=(1, port, golden_dir, out), nprocs=1, join=True)
    prob = np.load(out)
    assert np.array_equal(prob, tiny_io['prob_notta']) or float(np.abs(prob - tiny_io['prob_notta']).max()) <= 2e-6
