#!/usr/bin/env python
"""bench.py -- FLAIR volumes/sec of sliding-window inference (160x192x160, 128^3 patches, step 0.5,
Gaussian aggregation, fp32) on N MI355X.  One "step" = one whole volume through the hot path on
every rank (extract patches -> Generic_UNet forward -> seg head + softmax + mirror average + Gaussian
accumulate -> normalise + argmax), volume already resident in HBM.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: volumes are independent, so ranks shard the cohort (one volume per rank per step, no
data-path collective): weak scaling.  `--tta 0` drops nnU-Net's 8-fold mirroring (the reference's
`--disable_tta` stage-II predictions); the default keeps it, as `DeepWMH_predict` does.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FWD_GFLOP_PER_PATCH = 954.46
FWD_GBYTES_PER_PATCH = 4.17      # every layer reads its input once and writes its output once (fp32) + weights
AGG_GBYTES_PER_VOLUME = 0.46     # aggregation read-modify-write + volume in / label map out
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md


def cpu_baseline(plan, sd, vol, tta, gpu_patch_logits, budget_s=30.0):
    """Times the CPU oracle (kind 'port': our torch-CPU restatement of nnU-Net; the reference's own
    nnU-Net fork is not available) on a bounded sample and extrapolates to volumes/s."""
    import torch
    from oracle.generic_unet import unet_from_plan
    cores = torch.get_num_threads()                 # torch's default for this host (physical cores it may use)
    net = unet_from_plan(plan).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.do_ds = False
    P = plan.patch_size
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, :P[0], :P[1], :P[2]]))
    with torch.no_grad():
        t0 = time.time()
        net(x[:, :, :P[0] // 2, :P[1] // 2, :P[2] // 2])            # 1/8 of a patch forward: probe + warm-up
        t_probe = time.time() - t0
        if 8 * t_probe <= 4.0 * budget_s:       # the probe is cold (oneDNN primitive creation): generous
            frac, sample = 1.0, 'one %dx%dx%d patch forward (1/%d of the volume)' % (P[0], P[1], P[2], 64 if tta else 8)
            t0 = time.time()
            ref = net(x).numpy()
            t = time.time() - t0
        else:
            frac, sample, ref = 0.125, 'one %dx%dx%d sub-patch forward (1/8 patch), scaled x8' % (P[0] // 2, P[1] // 2, P[2] // 2), None
            t0 = time.time()
            net(x[:, :, :P[0] // 2, :P[1] // 2, :P[2] // 2])
            t = time.time() - t0
    forwards = 8 * (8 if tta else 1)
    est = forwards * t / frac
    out = dict(value=1.0 / est, unit='volumes/s', cores=cores, kind='port',
               sample='%s, %.1f s, torch-CPU fp32 oracle on %d threads; forward only (aggregation excluded)' % (sample, t, cores))
    parity = None
    if ref is not None and gpu_patch_logits is not None:
        def sm(a):
            e = np.exp(a - a.max(1, keepdims=True))
            return e / e.sum(1, keepdims=True)
        margin = np.abs(ref[:, 1] - ref[:, 0])
        parity = dict(sample='first patch, no mirroring',
                      max_abs_softmax_err=float(np.abs(sm(gpu_patch_logits) - sm(ref)).max()),
                      argmax_mismatches=int(((gpu_patch_logits.argmax(1) != ref.argmax(1)) & (margin > 4e-4)).sum()))
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--tta', type=int, default=1, help='1: 8-fold mirroring as DeepWMH_predict; 0: --disable_tta')
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['fp32', 'bf16'], default='fp32',
                    help='bf16 = BASELINE.json config 3 (bf16 matrix-core operands, fp32 accumulate/storage); not the headline number')
    ap.add_argument('--storage', choices=['fp32', 'bf16'], default='fp32',
                    help='bf16 (with --precision bf16): the full-resolution activation tensors are kept as bf16; not the headline number')
    ap.add_argument('--shard', choices=['volumes', 'patches'], default='volumes',
                    help='volumes: one volume per rank per step (weak scaling, no collective); patches: ONE volume per '
                         'step, its patches split over the ranks with an RCCL all-gather of the weighted tiles (strong)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)                  # before the process group: RCCL binds to the current device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)   # RCCL
    device = 'cuda:%d' % local_rank

    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    eng = UNetEngine(plan, sd, device, precision=args.precision, storage=args.storage)
    tta = bool(args.tta)
    pred = SlidingWindowPredictor(eng, do_mirroring=tta, batch=8)
    by_patch = args.shard == 'patches'
    vol_h = synthetic_flair(tuple(args.shape), seed=0 if by_patch else rank)
    vol = torch.from_numpy(vol_h).to(device)          # resident in HBM before the timed region
    if by_patch:
        from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
        sharded = PatchShardedPredictor(HipTileBackend(pred))
        run_step = lambda: sharded.predict(vol)
    else:
        run_step = lambda: pred.predict_device(vol, return_prob=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    eng.timed_cfgs = {0}                               # dominant kernel: conv3d_k3 config 0
    eng.timings = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seg, prob = run_step()
    barrier()
    dt = time.perf_counter() - t0
    eng.timed_cfgs = set()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant-kernel roofline from HIP events recorded on the launch stream during the timed region
    tot_ms = sum(e0.elapsed_time(e1) for (_, e0, e1, _, _) in eng.timings)
    tot_flop = sum(f for (_, _, _, f, _) in eng.timings)
    tot_bytes = sum(b for (_, _, _, _, b) in eng.timings)
    nl = max(1, len(eng.timings))
    achieved = tot_flop / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            traffic = json.load(f)['traffic_bytes_per_launch']     # PMC counters cannot be read live: separate rocprofv3 passes
    except Exception:
        pass
    roofline = dict(bound='mfma', kernel='conv3d_k3_ws_kernel<cfg0> (v_mfma_f32_32x32x2_f32 implicit GEMM, wave-specialised)',
                    achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic,
                    launches=len(eng.timings), avg_launch_ms=round(tot_ms / nl, 4),
                    algorithmic_gflop_per_launch=round(tot_flop / nl / 1e9, 3),
                    algorithmic_gbytes_per_launch=round(tot_bytes / nl / 1e9, 4),
                    share_of_step_time=round(tot_ms * 1e-3 / dt, 4))

    if rank == 0:
        forwards = 8 * (8 if tta else 1)
        nvol = args.steps if by_patch else args.gpus * args.steps
        res = dict(metric='FLAIR volumes/sec sliding-window inference (160x192x160)', value=nvol / dt,
                   unit='volumes/s', n_gpus=args.gpus, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling='strong' if by_patch else 'weak', vs_baseline=None,
                   dtype='f32' if args.precision == 'fp32' else ('bf16 operands / f32 accumulate+storage' if args.storage == 'fp32' else
                                                                 'bf16 operands, bf16 full-resolution tensors / f32 accumulate'), data='synthetic',
                   config=dict(workload='configs[1]: one %dx%dx%d FLAIR volume per GPU, 128^3 patches, step 0.5, fp32, '
                                        'Gaussian aggregation, mirroring (TTA x8) %s; %d patch forwards per volume'
                                        % (args.shape[0], args.shape[1], args.shape[2], 'on' if tta else 'off', forwards),
                               tta=tta, patch=[128, 128, 128], step=0.5, batch=8, weights='random He-init (seed 0)',
                               parallelism='patches of one volume sharded over ranks, RCCL all-gather of weighted tiles' if by_patch
                               else 'volumes sharded over ranks, no collective'),
                   end_to_end_tflops=round(forwards * FWD_GFLOP_PER_PATCH / 1e3 * nvol / dt, 2),
                   # north_star's "fraction of the HBM roofline" for the whole path (algorithmic fp32 bytes; the fp32 path is
                   # bound by the matrix cores, see roofline below and DESIGN.md 3.1)
                   hbm_whole_path=dict(algorithmic_gbytes_per_volume=round(forwards * FWD_GBYTES_PER_PATCH + AGG_GBYTES_PER_VOLUME, 2),
                                       achieved_gbs=round((forwards * FWD_GBYTES_PER_PATCH + AGG_GBYTES_PER_VOLUME) * nvol / dt, 1),
                                       peak_gbs=HBM_PEAK_GBS,
                                       frac=round((forwards * FWD_GBYTES_PER_PATCH + AGG_GBYTES_PER_VOLUME) * nvol / dt / HBM_PEAK_GBS, 4)),
                   roofline=roofline)
        if args.gpus == 1 and not args.no_cpu_baseline:
            # first patch, no mirroring, for the parity figure reported next to the baseline
            from deepwmh_amd import _lib
            P = plan.patch_size
            x = vol[None, :, :P[0], :P[1], :P[2]].contiguous()
            lg = eng.forward_logits(x).cpu().numpy()
            res['cpu_baseline'], res['parity'] = cpu_baseline(plan, sd, vol_h, tta, lg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
