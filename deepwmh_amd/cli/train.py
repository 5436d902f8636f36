"""`nnUNet_train` drop-in for the command lines DeepWMH emits
(/root/reference/deepwmh/pipeline/DCNN_multistage.py:292-309,501-520,693-713):
  nnUNet_train 3d_fullres nnUNetTrainerV2 <task id> all -e E -b B [--noval] [--save_every_epoch]
               [--disable_postprocessing_on_folds] [--custom_val_cases v1 v2 ...] [-c]
Writes model_latest / model_best / model_ep_NNNN (+ .pkl) and plans.pkl where the reference looks for them
(DCNN_multistage.py:300,321,510,608-631).  Launched under torch.distributed.run it is data parallel (RCCL
all-reduces of the gradient arena, started group by group while the backward pass is still running); the reference itself passes one GPU id."""
import argparse
import os
import pickle
import sys
import time


def build_parser():
    ap = argparse.ArgumentParser(prog='nnUNet_train')
    ap.add_argument('network')
    ap.add_argument('network_trainer')
    ap.add_argument('task')
    ap.add_argument('fold')
    ap.add_argument('-e', '--epochs', type=int, default=1000)                 # fork-only
    ap.add_argument('-b', '--batches_per_epoch', type=int, default=250)       # fork-only
    ap.add_argument('-val', '--validation_only', action='store_true')
    ap.add_argument('-c', '--continue_training', action='store_true')
    ap.add_argument('-p', default='nnUNetPlansv2.1')
    ap.add_argument('--noval', action='store_true')                           # fork-only
    ap.add_argument('--save_every_epoch', action='store_true')                # fork-only
    ap.add_argument('--disable_postprocessing_on_folds', action='store_true')
    ap.add_argument('--custom_val_cases', nargs='*', default=None)            # fork-only
    ap.add_argument('--npz', action='store_true')
    ap.add_argument('--fp32', action='store_true',                            # nnU-Net: disable mixed precision
                    help='exact-fp32 kernels throughout (also WMH_TRAIN_PRECISION=fp32).  The default is f16x2: fp32 tensors, statistics, '
                         'weight gradients and optimizer; forward and data-gradient products from fp16 pairs with a 2^16 loss scale -- '
                         'gradients at the fp32 step\'s level (nnU-Net itself defaults to fp16 AMP with a GradScaler)')
    ap.add_argument('--val_batches', type=int, default=None)
    ap.add_argument('--seed', type=int, default=12345)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.network != '3d_fullres' or args.network_trainer != 'nnUNetTrainerV2':
        sys.exit('nnUNet_train (deepwmh_amd): only "3d_fullres nnUNetTrainerV2" is implemented (DeepWMH uses nothing else)')
    results, pre = os.environ.get('RESULTS_FOLDER'), os.environ.get('nnUNet_preprocessed')
    if not results or not pre:
        sys.exit('RESULTS_FOLDER / nnUNet_preprocessed are not set')

    import numpy as np
    import torch
    import torch.distributed as dist
    from .. import checkpoint as ck
    from ..criterion import NUM_VAL_BATCHES_PER_EPOCH, ModelBestCriterion, epoch_metric
    from ..data import DATA_ID, PatchLoader, PrefetchLoader
    from ..plan import UNetPlan
    from ..synthetic import he_init_state_dict
    from ..trainer import UNetTrainer, downsample_targets, poly_lr

    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        sys.exit('nnUNet_train (deepwmh_amd): no HIP device visible')
    # WMH_TRAIN_BACKEND=gloo: ranks may share GPUs (rank r on device r mod the visible count; gradients averaged through gloo) -- how the
    # data-parallel command is exercised where fewer GPUs than ranks exist (tests/test_gpu_two_ranks.py); the default, nccl = RCCL, wants one each
    backend = os.environ.get('WMH_TRAIN_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)                        # before the process group: RCCL binds to the current device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
    device = 'cuda:%d' % local

    tasks = [d for d in sorted(os.listdir(pre)) if d.startswith('Task%03d_' % int(args.task))] if args.task.isdigit() else [args.task]
    if len(tasks) != 1 or not os.path.isdir(os.path.join(pre, tasks[0])):
        sys.exit('task %s not found under %s (run nnUNet_plan_and_preprocess first)' % (args.task, pre))
    task = tasks[0]
    with open(os.path.join(pre, task, '%s_plans_3D.pkl' % args.p), 'rb') as f:
        plans = pickle.load(f)
    plan = UNetPlan.from_nnunet_plans(plans)
    stage_dir = os.path.join(pre, task, DATA_ID + '_stage0')
    cases = sorted(f[:-4] for f in os.listdir(stage_dir) if f.endswith('.npz'))
    val_cases = [c for c in (args.custom_val_cases or []) if c]
    missing = [c for c in val_cases if c not in cases]
    if missing:
        sys.exit('custom validation cases not in the dataset: %s' % ' '.join(missing))
    train_cases = [c for c in cases if c not in val_cases] if val_cases else cases     # fold "all": train on everything
    val_cases = val_cases or cases
    folder = ck.trainer_folder(results, task, args.network, args.network_trainer, args.p)
    fdir = os.path.join(folder, ck.fold_folder(args.fold))
    os.makedirs(fdir, exist_ok=True)

    bs = int(plans['plans_per_stage'][max(plans['plans_per_stage'].keys())]['batch_size'])
    precision = os.environ.get('WMH_TRAIN_PRECISION', 'f16x2')
    if args.fp32:
        precision = 'fp32'
    if precision not in ('fp32', 'f16x2'):              # (the bf16 mixed-precision mode of rounds 1-5 -- gradients 8.6 % off -- is gone)
        sys.exit('nnUNet_train (deepwmh_amd): WMH_TRAIN_PRECISION must be fp32 or f16x2, got %r' % precision)
    start_epoch, best, ma_resume = 0, None, None
    latest = os.path.join(fdir, 'model_latest.model')
    if args.continue_training and os.path.isfile(latest):
        _, _, sd, info = ck.load_checkpoint(folder, args.fold, 'model_latest')
        start_epoch = int(info['epoch'])
        tr = UNetTrainer(plan, sd, device, precision=precision)
        mom = info['checkpoint'].get('deepwmh_amd_momentum')
        if mom is not None:
            tr.mom.copy_(torch.as_tensor(np.asarray(mom)).to(tr.mom.device))
            tr.first_step = False
        # the loss scale belongs to the precision it was found in: a checkpoint of an fp32 run stores 1.0, and continuing it with
        # the f16x2 backward at scale 1 would run fp16-pair gradients in the regime trainer.py documents as 7 % off
        saved_scale, saved_prec = info['checkpoint'].get('deepwmh_amd_loss_scale'), info['checkpoint'].get('deepwmh_amd_precision')
        if saved_scale and tr.loss_scale != 1.0 and saved_prec == precision:
            tr.loss_scale = float(saved_scale)
            if rank == 0:
                print('loss scale %g restored from the checkpoint (%s)' % (tr.loss_scale, precision), flush=True)
        elif saved_scale and rank == 0 and tr.loss_scale != 1.0:
            print('checkpoint was written by a %s run: its loss scale %g is not restored, starting at %g'
                  % (saved_prec or 'pre-round-5', float(saved_scale), tr.loss_scale), flush=True)
        best = info['checkpoint'].get('deepwmh_amd_best')
        ma_resume = info['checkpoint'].get('deepwmh_amd_ma')
        if rank == 0:
            print('continuing from %s (epoch %d)' % (latest, start_epoch), flush=True)
    else:
        tr = UNetTrainer(plan, he_init_state_dict(plan, seed=args.seed, trained_like=False), device, precision=precision)
    if rank == 0:
        print('deepwmh_amd nnUNet_train: precision %s, %d rank(s)' % (precision, world), flush=True)
    if world > 1:                                       # identical start on every rank
        dist.broadcast(tr.params, 0); dist.broadcast(tr.mom, 0)
    loader = PrefetchLoader(PatchLoader(stage_dir, train_cases, plan.patch_size, bs, seed=args.seed + 1000 * rank + start_epoch))
    vloader = PatchLoader(stage_dir, val_cases, plan.patch_size, bs, seed=args.seed + 77, mirror=False)
    nval = args.val_batches if args.val_batches is not None else NUM_VAL_BATCHES_PER_EPOCH      # nnU-Net: 50, whatever -b says
    crit = ModelBestCriterion(ma_resume, best)          # nnU-Net's rule (criterion.py); the moving average continues across `-c`

    # Checkpoint files are written by a helper PROCESS, in order (cli/_ckpt_writer.py says why not a thread): the training process only takes
    # the host snapshot -- one D2H copy of the flat parameter arena per epoch (and of the momentum where training can continue from), one raw
    # file write each -- and goes on launching kernels.
    writer = None
    if rank == 0:
        offs = [tr.grad_ranges[name][0] for name, _ in tr.param_order]
        writer = ck.CheckpointProcess(tr.param_order, offs, plans, fdir)
    snap = dict(epoch=None)

    def save(name, epoch):
        if rank != 0:
            return
        if snap['epoch'] != epoch:                           # parameters as they are now; later files of this epoch reuse the snapshot
            snap['epoch'] = epoch                            # (snapshot() itself waits until at most one older snapshot is alive)
            writer.snapshot(tr.params.detach().cpu().numpy())
        # nnU-Net's checkpoint carries plain types; the json line does too (best / ma are floats or None)
        extra = dict(deepwmh_amd_best=None if best is None else float(best), deepwmh_amd_ma=None if ma is None else float(ma),
                     deepwmh_amd_loss_scale=float(tr.loss_scale), deepwmh_amd_precision=precision)
        latest = name == 'model_latest'                      # the SGD momentum (as large as the weights) only where training can continue from
        if latest:
            writer.add_momentum(tr.mom.detach().cpu().numpy())
        writer.save(folder, args.fold, name, epoch, extra, with_momentum=latest)

    ma = ma_resume
    try:
        for epoch in range(start_epoch, args.epochs):
            lr = poly_lr(epoch, args.epochs, 1e-2, 0.9)
            t0, losses = time.time(), []
            for _ in range(args.batches_per_epoch):
                data, seg = loader.next_batch()
                x = torch.from_numpy(data).to(device)
                tg = downsample_targets(torch.from_numpy(seg).to(device), plan.num_pool, plan.pool_op_kernel_sizes)
                losses.append(tr.run_iteration(x, tg, lr).clone())           # stays on the device: no per-iteration sync
            losses = [float(v) for v in torch.stack(losses).cpu()]          # synchronises: the epoch's iterations are done
            t_train = time.time() - t0
            skipped_before, losses = getattr(tr, '_skipped_reported', 0), [v for v in losses if np.isfinite(v)] or [float('nan')]
            vl = []
            if getattr(tr, 'eval_counts', None) is not None:
                tr.eval_counts.zero_()
            for _ in range(nval):
                data, seg = vloader.next_batch()
                x = torch.from_numpy(data).to(device)
                tg = downsample_targets(torch.from_numpy(seg).to(device), plan.num_pool, plan.pool_op_kernel_sizes)
                vl.append(tr.run_iteration(x, tg, lr, do_backprop=False).clone())
                tr.online_eval(tg[0])                             # hard tp / fp / fn of the full-resolution output, summed over the epoch's batches
            vl = [float(t) for t in torch.stack(vl).cpu()]
            v = float(np.mean(vl))
            metric = epoch_metric(tr.eval_counts.cpu().numpy(), plan.num_classes)
            is_best = crit.update(metric, v)
            ma, best = crit.ma, crit.best
            if rank == 0:
                print('epoch %d/%d: lr %.6f train loss %.4f val loss %.4f val dice %s MA %.4f%s (%.1f s: %d iterations at %.1f ms, validation %.2f s)'
                      % (epoch + 1, args.epochs, lr, float(np.mean(losses)), v, 'n/a' if metric is None else '%.4f' % metric, ma,
                         ' -> model_best' if is_best else '', time.time() - t0, args.batches_per_epoch,
                         1e3 * t_train / max(1, args.batches_per_epoch), time.time() - t0 - t_train), flush=True)
                if tr.skipped_steps != skipped_before:      # GradScaler semantics: the device skipped updates whose gradient norm was not finite
                    print('  %d optimizer step(s) skipped (non-finite gradients); loss scale now %g' % (tr.skipped_steps - skipped_before, tr.loss_scale), flush=True)
            tr._skipped_reported = tr.skipped_steps
            if is_best:
                save('model_best', epoch + 1)
            if args.save_every_epoch:
                save('model_ep_%04d' % (epoch + 1), epoch + 1)
            save('model_latest', epoch + 1)                  # last: carries this epoch's `best`, so a `-c` run cannot overwrite a better model_best
        save('model_final_checkpoint', args.epochs)
    except BaseException:
        # (an exception, Ctrl-C or sys.exit inside the loop: the helper process and its several hundred MB of raw arenas in /dev/shm must not stay behind)
        if writer is not None:
            writer.abort()
        loader.close()
        raise
    if writer is not None:
        writer.close()                                       # waits for every file; a failed write fails the command
    loader.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
