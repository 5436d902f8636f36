"""What ONE rank of the two multi-GPU configurations does, measured on one MI355X, and the run it predicts (no 8-GPU node
has been available to the build, so the multi-GPU figures in the documentation are these predictions, labelled as such).

configs[3] (patches of one volume over W ranks): per round a rank computes ONE tile -- its 8 mirrored copies are a batch-8 forward -- into a
Gaussian-weighted 16.8 MB softmax tile, all ranks exchange their tiles (ONE all_gather_into_tensor of W x 16.8 MB, asynchronous, overlapped with
the next round's forward when there is one) and every rank accumulates the W tiles in patch order.  Measured here: the tile (forward + head),
the RCCL call on a one-rank group (launch + copy: what is NOT wire time), the accumulate of one tile, the finalize.  Wire time is modelled: a
ring all-gather moves (W - 1) x 16.8 MB per rank over one xGMI link at `link_gbs` GB/s (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, a ring
is bound by one link).
configs[4] (data-parallel training): the iteration measured on one GPU + the part of the 124.7 MB gradient all-reduce (ring: 2 (W - 1) / W x the
bytes over one link) that the backward pass does not cover -- the arena is sent in runs as the backward pass completes them, the last run (the first
encoder layers, small) after it -- + the CUs a resident RCCL kernel takes from the persistent grids while it runs.
    python tools/shard_cost.py profiles/rNN_shard_cost.json"""
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)          # RCCL, before anything else touches the GPU
from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor  # noqa: E402
from deepwmh_amd.network import UNetEngine  # noqa: E402
from deepwmh_amd.plan import baseline_plan  # noqa: E402
from deepwmh_amd.sliding_window import SlidingWindowPredictor  # noqa: E402
from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair  # noqa: E402
from deepwmh_amd.trainer import UNetTrainer, downsample_targets  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


plan = baseline_plan()
sd = he_init_state_dict(plan, seed=0)
eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2')
vol = torch.from_numpy(synthetic_flair((160, 192, 160), seed=3)).cuda()
out = dict(how='tools/shard_cost.py on one MI355X; wire times are modelled, everything else is measured', link_gbs_assumed=[100.0, 153.0])

# ---- configs[3] ------------------------------------------------------------------------------------------------------------------
pred8 = SlidingWindowPredictor(eng, do_mirroring=True, batch=8)
pred16 = SlidingWindowPredictor(eng, do_mirroring=True, batch=16)
be = HipTileBackend(pred8)
A, pad_lo, origins, K, P = be.geometry(tuple(vol.shape[1:]))
ms_tile = timed(lambda: be.tile(vol, origins[3]), 6)
tile = be.tile(vol, origins[3])
agg = be.zeros((K,) + A)
ms_acc = timed(lambda: be.accumulate(agg, tile, origins[3]), 20)
cnt = be.count_map(A, origins)
ms_fin = timed(lambda: be.finalize(agg, cnt, pad_lo, tuple(vol.shape[1:])), 20)
send, recv = be.zeros((K,) + tuple(P)), be.zeros((1, K) + tuple(P))
ms_gather_call = timed(lambda: dist.all_gather_into_tensor(recv.view((K,) + tuple(P)), send), 20)
ms_vol_1gpu_b16 = timed(lambda: pred16.predict_device(vol, return_prob=True), 3)
ms_vol_1gpu_b8 = timed(lambda: pred8.predict_device(vol, return_prob=True), 3)
ms_vol_sharded_1rank = timed(lambda: PatchShardedPredictor(be, force_collective=True).predict(vol), 3)
tile_mb = K * int(np.prod(P)) * 4 / 1e6
c3 = dict(tile_ms_batch8_forward_plus_head=round(ms_tile, 3), accumulate_one_tile_ms=round(ms_acc, 3), finalize_ms=round(ms_fin, 3),
          all_gather_call_one_rank_ms=round(ms_gather_call, 3), tile_megabytes=round(tile_mb, 2),
          whole_volume_one_gpu_ms=dict(batch16=round(ms_vol_1gpu_b16, 2), batch8=round(ms_vol_1gpu_b8, 2), sharded_path_one_rank_forced_collective=round(ms_vol_sharded_1rank, 2)),
          predicted={})
ntiles = len(origins)
for W in (2, 4, 8):
    rounds = -(-ntiles // W)
    row = {}
    for link in out['link_gbs_assumed']:
        wire = (W - 1) * tile_mb / link                        # ms: MB / (GB/s) = ms
        # every round: the tile; the exchange of round r overlaps with the tile of round r + 1, the last one is exposed; W accumulates per round
        exposed = wire + ms_gather_call
        per_round_hidden = max(0.0, wire + ms_gather_call - ms_tile)
        ms = rounds * ms_tile + (rounds - 1) * per_round_hidden + exposed + ntiles * ms_acc + ms_fin
        row['link_%d_gbs' % link] = dict(ms_per_volume=round(ms, 2), volumes_per_s=round(1e3 / ms, 2),
                                         strong_scaling_efficiency_vs_one_gpu_batch16=round((ms_vol_1gpu_b16 / ms) / W, 3))
    c3['predicted']['W%d' % W] = dict(rounds=rounds, **row)
c3['note'] = ('patch sharding is the LATENCY mode (one volume as fast as possible); for throughput the cohort is sharded by volume (no collective): W x the one-GPU rate '
              'minus nothing but host effects')
out['configs3_patch_sharded'] = c3
del pred8, pred16, be, agg, tile
torch.cuda.empty_cache()

# ---- configs[4] ------------------------------------------------------------------------------------------------------------------
tr = UNetTrainer(plan, sd, 'cuda:0', precision='f16x2')
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 1, 128, 128, 128, generator=g).cuda()
lab = (torch.rand(2, 128, 128, 128, generator=g) < 0.05).to(torch.uint8).cuda()
td = downsample_targets(lab, plan.num_pool, plan.pool_op_kernel_sizes)
ms_iter = timed(lambda: tr.run_iteration(x, td, 1e-2), 8)
def forced_iteration():                                        # the backward pass with its all-reduces issued through the one-rank RCCL group
    tr.forward(x)
    loss = tr.backward(td, None, force_collective=True)
    tr.optimizer_step(1e-2)
    return loss


ms_iter_forced = timed(forced_iteration, 8)
ms_fwd = timed(lambda: tr.run_iteration(x, td, 1e-2, do_backprop=False), 8)
grad_mb = tr.nparams * 4 / 1e6
c4 = dict(iteration_ms_one_gpu=round(ms_iter, 3), iteration_ms_one_rank_group=None if ms_iter_forced is None else round(ms_iter_forced, 3),
          forward_and_loss_ms=round(ms_fwd, 3), gradient_megabytes=round(grad_mb, 1), predicted={})
bwd_ms = ms_iter - ms_fwd
for W in (2, 4, 8):
    row = {}
    for link in out['link_gbs_assumed']:
        wire = 2.0 * (W - 1) / W * grad_mb / link              # ring all-reduce, one link
        # the runs are issued while the backward pass runs (the deep stages' 100 MB are final after ~40 % of it); what cannot overlap: the last run
        # (first encoder layers + full-resolution decoder, ~1 MB) and whatever of the wire time exceeds the rest of the backward pass
        exposed = max(0.0, wire - 0.6 * bwd_ms) + 0.15
        steal = wire * 0.06                                      # a resident RCCL kernel holds ~16 of the 256 CUs while it runs
        ms = ms_iter + exposed + steal
        row['link_%d_gbs' % link] = dict(all_reduce_wire_ms=round(wire, 2), iteration_ms=round(ms, 2), patches_per_s=round(2 * W * 1e3 / ms, 1),
                                         weak_scaling_efficiency=round(ms_iter / ms, 3))
    c4['predicted']['W%d' % W] = row
out['configs4_data_parallel_training'] = c4
dist.destroy_process_group()
# (RCCL prints a version banner to stdout: the record goes to the file named on the command line, or behind the banner)
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(out, f, indent=1)
else:
    print(json.dumps(out, indent=1))
