"""`nnUNet_predict` drop-in: the command DeepWMH shells out to for every prediction.

Accepts exactly the flag set the reference emits
(/root/reference/deepwmh/main/predict.py:153-155;
 /root/reference/deepwmh/pipeline/DCNN_multistage.py:331-336,531-533):
  -i IN -o OUT -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t TASK -f all -chk NAME
  --disable_post_processing [--disable_tta] [--save_softmax] --selected_cases c1 c2 ...
and the environment contract RESULTS_FOLDER / CUDA_VISIBLE_DEVICES (predict.py:101,150-152).
`--selected_cases` and `--save_softmax` follow the fork's semantics: only the listed cases are predicted;
`--save_softmax` additionally writes channel 0 (background probability) as <case>_0.nii.gz
(DCNN_multistage.py:340-344,359).  Errors exit non-zero (external_call.py:64-72 aborts the pipeline).

Multi-GPU extension (the reference passes exactly one GPU id, predict.py:150): with
`MI355WMH_GPUS=0-7` (or `0,2,5`, or `all`) in the environment this process does not touch a GPU at all.  It
starts one child `nnUNet_predict` per GPU with `CUDA_VISIBLE_DEVICES=<id>` and nnU-Net's own `--part_id r
--num_parts N`, so that the cases are sharded over the GPUs with no data-path collective, waits, and exits
non-zero if any child did.
"""
import argparse
import os
import subprocess
import sys
import time


def build_parser():
    ap = argparse.ArgumentParser(prog='nnUNet_predict', description='MI355X-native nnU-Net 3d_fullres prediction for DeepWMH')
    ap.add_argument('-i', '--input_folder', required=True)
    ap.add_argument('-o', '--output_folder', required=True)
    ap.add_argument('-t', '--task_name', required=True)
    ap.add_argument('-tr', '--trainer_class_name', default='nnUNetTrainerV2')
    ap.add_argument('-ctr', '--cascade_trainer_class_name', default=None)
    ap.add_argument('-m', '--model', default='3d_fullres')
    ap.add_argument('-p', '--plans_identifier', default='nnUNetPlansv2.1')
    ap.add_argument('-f', '--folds', nargs='+', default=['all'])
    ap.add_argument('-z', '--save_npz', action='store_true')
    ap.add_argument('-l', '--lowres_segmentations', default='None')
    ap.add_argument('--part_id', type=int, default=0)
    ap.add_argument('--num_parts', type=int, default=1)
    ap.add_argument('--num_threads_preprocessing', type=int, default=6)
    ap.add_argument('--num_threads_nifti_save', type=int, default=2)
    ap.add_argument('--disable_tta', action='store_true')
    ap.add_argument('--overwrite_existing', action='store_true')
    ap.add_argument('--mode', default='normal')
    ap.add_argument('--all_in_gpu', default='None')
    ap.add_argument('--step_size', type=float, default=0.5)
    ap.add_argument('-chk', default='model_final_checkpoint')
    ap.add_argument('--disable_mixed_precision', action='store_true')
    ap.add_argument('--disable_post_processing', action='store_true')
    ap.add_argument('--selected_cases', nargs='*', default=None)      # fork-only
    ap.add_argument('--save_softmax', action='store_true')            # fork-only
    return ap


def _env_id_list(var):
    vis = os.environ.get(var)
    if vis is None or vis.strip() == '':
        return None
    return [v.strip() for v in vis.split(',') if v.strip()]


def count_gpus_without_runtime(topology='/sys/class/kfd/kfd/topology/nodes', probe_cmd=None, probe_timeout=120):
    """Number of GPUs a child of this process would see, found WITHOUT initialising the HIP runtime here (the fan-out parent only
    starts children).  A short-lived child asks the runtime itself (`torch.cuda.device_count()`), which sees what the container's
    cgroup and ROCR_VISIBLE_DEVICES leave visible -- with HIP_/CUDA_VISIBLE_DEVICES removed, exactly as fan_out's children run (the
    reference sets CUDA_VISIBLE_DEVICES to ONE id, predict.py:150-152; MI355WMH_GPUS replaces it).  Only if that child cannot be
    started or does not answer within `probe_timeout` seconds: the KFD topology in sysfs (nodes with simd_count > 0; CPU nodes have 0)
    narrowed by ROCR_VISIBLE_DEVICES -- which counts every GPU of the HOST, so it is accepted only when it cannot overcount: on a host
    whose device files (/dev/dri/renderD*) are as many as the topology's GPUs.  Otherwise: fail and ask for explicit ids."""
    cmd = probe_cmd if probe_cmd is not None else [sys.executable, '-c', 'import torch; print(torch.cuda.device_count())']
    why = 'no probe command'
    if cmd:
        try:
            env = {k: v for k, v in os.environ.items() if k not in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')}
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=probe_timeout, env=env).stdout.decode().split()
            n = int(out[-1])
            if n > 0:
                print('deepwmh_amd nnUNet_predict: MI355WMH_GPUS=all -> %d GPU(s) (counted by the HIP runtime in a child process)' % n, flush=True)
                return n
            why = 'the runtime probe saw no GPU'
        except Exception as e:
            why = 'the runtime probe failed (%s)' % type(e).__name__
    n = 0
    try:
        nodes = sorted(os.listdir(topology), key=lambda v: int(v) if v.isdigit() else 0)
    except OSError:
        nodes = []
    for node in nodes:
        try:
            with open(os.path.join(topology, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) > 0:
            n += 1
    ids = _env_id_list('ROCR_VISIBLE_DEVICES')
    if ids is not None:
        n = min(n, len(ids))
    elif topology == '/sys/class/kfd/kfd/topology/nodes':       # a restricted container exposes fewer render nodes than the host has GPUs
        try:
            nrender = len([f for f in os.listdir('/dev/dri') if f.startswith('renderD')])
        except OSError:
            nrender = 0
        if nrender < n:
            raise RuntimeError('MI355WMH_GPUS=all: %s, and sysfs lists %d GPU(s) of the host while %d device file(s) are visible here; '
                               'list the ids explicitly (e.g. MI355WMH_GPUS=0-7)' % (why, n, nrender))
    if n < 1:
        raise RuntimeError('MI355WMH_GPUS=all: no GPU visible (%s, %s empty); list the ids explicitly '
                           '(e.g. MI355WMH_GPUS=0-7)' % (why, topology))
    print('deepwmh_amd nnUNet_predict: MI355WMH_GPUS=all -> %d GPU(s) (%s; counted from %s)' % (n, why, topology), flush=True)
    return n


def parse_gpu_list(spec):
    """'0-7' -> [0..7]; '0,2,5' -> [0,2,5]; '1' -> [1]; 'all' -> every visible device; ''/None -> [].  An id may repeat
    ('0,0'): that many processes then share the GPU."""
    if spec is None or not spec.strip():
        return []
    spec = spec.strip()
    if spec == 'all':
        return list(range(count_gpus_without_runtime()))
    ids = []
    for part in spec.split(','):
        part = part.strip()
        if '-' in part:
            lo, hi = part.split('-')
            lo, hi = int(lo), int(hi)
            if hi < lo:
                raise ValueError('MI355WMH_GPUS: empty range %r' % part)
            ids += list(range(lo, hi + 1))
        elif part:
            ids.append(int(part))
    if any(i < 0 for i in ids):
        raise ValueError('MI355WMH_GPUS: negative GPU id in %r' % spec)
    return ids


def list_cases(input_folder, selected):
    """Case ids of the input folder (files <case>_0000.nii.gz), restricted to --selected_cases (fork-only flag; the
    reference pads the list with extra spaces, DCNN_multistage.py:327-333, hence the empty-string filter)."""
    files = sorted(f for f in os.listdir(input_folder) if f.endswith('_0000.nii.gz'))
    cases = [f[:-len('_0000.nii.gz')] for f in files]
    if selected:
        wanted = [c for c in selected if c]
        missing = [c for c in wanted if c not in cases]
        if missing:
            sys.exit('selected cases not found in %s: %s' % (input_folder, ' '.join(missing)))
        cases = [c for c in cases if c in wanted]
    return cases


def fan_out(argv, gpus, ncases, child_cmd=None, poll_s=0.05):
    """Parent of a multi-GPU prediction: one child per GPU, case sharding by --part_id/--num_parts.  The parent never
    initialises a GPU.  Returns the exit code: 0 if every child returned 0, else the first failing child's code (the
    other children are terminated -- the reference aborts its pipeline on the first failed command,
    /root/reference/deepwmh/utilities/external_call.py:64-72)."""
    n = max(1, min(len(gpus), ncases))
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    child_cmd = child_cmd or [sys.executable, '-m', 'deepwmh_amd.cli.predict']
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.pop('MI355WMH_GPUS', None)                          # the children are plain single-GPU runs
        # the reference's own convention (predict.py:150-152): one id in CUDA_VISIBLE_DEVICES.  The reference itself hands this command
        # CUDA_VISIBLE_DEVICES=<one id>, which MI355WMH_GPUS exists to widen: its ids replace that variable (and HIP_VISIBLE_DEVICES);
        # ROCR_VISIBLE_DEVICES, the layer below HIP, is inherited, so the ids count within what the container was given
        env['CUDA_VISIBLE_DEVICES'] = str(gpus[r])
        env.pop('HIP_VISIBLE_DEVICES', None)
        env['PYTHONPATH'] = root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        cmd = list(child_cmd) + list(argv) + ['--part_id', str(r), '--num_parts', str(n)]
        procs.append((r, gpus[r], subprocess.Popen(cmd, env=env)))
    print('deepwmh_amd nnUNet_predict: %d case(s) over %d GPU(s) %s' % (ncases, n, [g for _, g, _ in procs]), flush=True)
    rc_final, live = 0, list(procs)
    while live:
        for item in list(live):
            r, g, p = item
            rc = p.poll()
            if rc is None:
                continue
            live.remove(item)
            if rc != 0 and rc_final == 0:
                rc_final = rc if rc > 0 else 1
                print('deepwmh_amd nnUNet_predict: part %d on GPU %d exited with code %d (its cases, indices %s of the sorted case '
                      'list, are not predicted); stopping the other parts' % (r, g, rc, list(range(ncases))[r::n]), file=sys.stderr, flush=True)
                for _, _, q in live:
                    q.terminate()
        if live:
            time.sleep(poll_s)
    return rc_final


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = build_parser().parse_args(argv)
    if args.model != '3d_fullres':
        sys.exit('nnUNet_predict (deepwmh_amd): only -m 3d_fullres is implemented (DeepWMH uses nothing else)')
    if len(args.folds) != 1:
        sys.exit('nnUNet_predict (deepwmh_amd): exactly one fold (-f all) is supported')
    results = os.environ.get('RESULTS_FOLDER')
    if not results:
        sys.exit('RESULTS_FOLDER is not set')
    gpus = parse_gpu_list(os.environ.get('MI355WMH_GPUS'))
    if len(gpus) > 1 and args.num_parts == 1:
        return fan_out(argv, gpus, len(list_cases(args.input_folder, args.selected_cases)))
    if len(gpus) == 1 and args.num_parts == 1:
        os.environ['CUDA_VISIBLE_DEVICES'] = str(gpus[0])       # before torch initialises the runtime

    import numpy as np
    import torch
    from .. import nifti
    from ..checkpoint import load_checkpoint, resolve_task, trainer_folder
    from ..network import UNetEngine
    from ..sliding_window import SlidingWindowPredictor

    task = resolve_task(results, args.task_name, args.model)
    folder = trainer_folder(results, task, args.model, args.trainer_class_name, args.plans_identifier)
    plan, plans, sd, info = load_checkpoint(folder, args.folds[0], args.chk)
    if not torch.cuda.is_available():
        sys.exit('nnUNet_predict (deepwmh_amd): no HIP device visible (CUDA_VISIBLE_DEVICES=%s)'
                 % os.environ.get('CUDA_VISIBLE_DEVICES'))
    # nnU-Net predicts under fp16 autocast unless --disable_mixed_precision is given.  Here the default is 'f16x2': fp32 tensors,
    # fp32 accumulation, every product formed from fp16 pairs (21-22 significant bits per operand) -- softmax within 1e-5 of the
    # fp32 CPU oracle at full size (the parity bar is 1e-4; tests/test_gpu_network.py), 1.9x the exact-fp32 kernels' speed.
    # --disable_mixed_precision (or WMH_PREDICT_PRECISION=fp32) selects the exact-fp32 kernels.  Reduced precision stays opt-in:
    # WMH_PREDICT_PRECISION=bf16 (bf16 matrix-core operands), WMH_PREDICT_STORAGE=bf16 (additionally channels-last bf16 tensors).
    precision = os.environ.get('WMH_PREDICT_PRECISION', 'f16x2')
    storage = os.environ.get('WMH_PREDICT_STORAGE', 'fp32')
    if args.disable_mixed_precision:
        precision, storage = 'fp32', 'fp32'
    eng = UNetEngine(plan, sd, 'cuda:0', precision=precision, storage=storage)   # CUDA_VISIBLE_DEVICES already selected the GPU (predict.py:150)
    pred = SlidingWindowPredictor(eng, plan.patch_size, args.step_size, True, not args.disable_tta, batch=16)   # two tiles' mirrored copies per forward

    cases = list_cases(args.input_folder, args.selected_cases)[args.part_id::args.num_parts]
    os.makedirs(args.output_folder, exist_ok=True)
    nmod = plan.num_input_channels
    print('deepwmh_amd nnUNet_predict: task %s, checkpoint %s (epoch %s), %d case(s), mirroring %s'
          % (task, args.chk, info.get('epoch'), len(cases), 'off' if args.disable_tta else 'on'), flush=True)
    # Host work (gunzip + NIfTI parse before, gzip + write after) runs in worker threads around the GPU, as nnU-Net's
    # --num_threads_preprocessing / --num_threads_nifti_save processes do; zlib and numpy release the GIL.
    from concurrent.futures import ThreadPoolExecutor
    from ..preprocess import export_prediction_device, preprocess_case_device
    stage = info['init'][5] if info.get('init') else None
    todo = []
    for case in cases:
        out = os.path.join(args.output_folder, case + '.nii.gz')
        if os.path.isfile(out) and not args.overwrite_existing and not args.save_softmax:
            continue
        todo.append((case, out))

    # Host buffers around the GPU are page-locked and recycled (round 6): the readers decompress straight into them, the H2D copy of a case
    # and the D2H copies of its label map / probabilities are stream-ordered DMA instead of `.to()` / `.cpu()`'s pageable staging -- without
    # mirroring a case is 22 ms of GPU work, and the two pageable copies of a 20 MB float volume were a third of that again on the lane's
    # thread.  WMH_PREDICT_PINNED=0 restores the pageable path.
    import threading
    use_pinned = os.environ.get('WMH_PREDICT_PINNED', '1') != '0'

    class PinnedPool:
        def __init__(self):
            self.free, self.lock = {}, threading.Lock()

        def get(self, shape, dtype):
            key = (tuple(int(v) for v in shape), dtype)
            with self.lock:
                lst = self.free.get(key)
                if lst:
                    return lst.pop()
            return torch.empty(key[0], dtype=dtype).pin_memory()

        def put(self, t):
            if t is not None:
                with self.lock:
                    self.free.setdefault((tuple(t.shape), t.dtype), []).append(t)
    pool = PinnedPool()
    tsum = dict(read=0.0, wait=0.0, pre=0.0, predict=0.0, export=0.0, write=0.0)      # seconds summed over cases (lanes / threads run side by side)
    tlock = threading.Lock()

    def load(case):
        t0 = time.time()
        vols, meta, held = [], None, [None]
        for m in range(nmod):
            path = os.path.join(args.input_folder, '%s_%04d.nii.gz' % (case, m))
            if use_pinned:
                def alloc(shape, m=m):
                    if held[0] is None:
                        held[0] = pool.get((nmod,) + tuple(shape), torch.float32)
                    return held[0][m].numpy()
                arr, mt = nifti.read(path, alloc=alloc)
            else:
                arr, mt = nifti.read(path)
            vols.append(arr)
            meta = meta or mt
        if held[0] is not None and all(v.shape == tuple(held[0].shape[1:]) for v in vols):
            vol = held[0]                                       # a page-locked tensor [C, z, y, x]: every modality was read into its slice
        else:
            pool.put(held[0]); held[0] = None
            vol = np.stack(vols)
        dt = time.time() - t0
        with tlock:
            tsum['read'] += dt
        return vol, meta, dt

    def save(out, case, seg_full, prob0, meta, held=()):
        t0 = time.time()
        try:
            nifti.write(out, seg_full, like=meta, dtype=np.uint8)
            if prob0 is not None:
                nifti.write(os.path.join(args.output_folder, case + '_0.nii.gz'), prob0, like=meta, dtype=np.float32)
        finally:
            for t in held:
                pool.put(t)
            with tlock:
                tsum['write'] += time.time() - t0

    # gunzip of one float32 case is ~0.2 s of one core and cannot be split, so the readers bound a --disable_tta run (0.03 s of GPU per case):
    # as many as --num_threads_preprocessing allows (nnU-Net's default is 6)
    # (round 6: with nnU-Net's default of 6 the readers delivered 30 cases/s at best and the command ran at 27.6 -- the readers were the bound,
    #  not the GPU (45 cases/s); on hosts with cores to spare their number follows the host, up to 16)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    nread = max(1, int(os.environ.get('WMH_PREDICT_READERS', min(max(args.num_threads_preprocessing, ncpu // 8), 16))))
    # Cases in flight: without mirroring a case is 25 ms of GPU work between two synchronising copies (the volume in, label map and
    # probabilities out) and the host work around them -- a second case on a second stream, with its own activation buffers on the same
    # weights, fills those gaps (the Stage-II predictions, DCNN_multistage.py:331-344: 5 x N cases with --disable_tta --save_softmax).
    # With mirroring the GPU is busy 94 % of the time already and a second set of buffers (29 GB, half a second to allocate) does not pay.
    ninflight = max(1, int(os.environ.get('WMH_PREDICT_INFLIGHT', '2' if args.disable_tta else '1')))
    lanes = [(pred, torch.cuda.current_stream())]
    for _ in range(1, min(ninflight, max(1, len(todo)))):
        e2 = eng.clone()
        lanes.append((SlidingWindowPredictor(e2, plan.patch_size, args.step_size, True, not args.disable_tta, batch=16), torch.cuda.Stream()))
    lock = threading.Lock()
    # (the lanes' threads hold the GIL for a millisecond at a time -- ctypes launches -- and hand it to each other, the readers and the writers:
    #  with CPython's default 5 ms switch interval a lane waiting for the GIL loses a fifth of a case's GPU time per hand-over)
    sys.setswitchinterval(float(os.environ.get('WMH_PREDICT_SWITCH_S', '5e-4')))
    t_all = time.time()
    t_case_done = []
    # the label map is small; the background probability (--save_softmax) is a float32 volume: its compression is spread over the gzip pool
    # (nifti.gzip_members), so a few writer threads keep up with 40 cases a second
    nwrite = max(1, args.num_threads_nifti_save, 4 if args.save_softmax else 1)
    with ThreadPoolExecutor(nread) as readers, ThreadPoolExecutor(nwrite) as writers, ThreadPoolExecutor(len(lanes)) as gpu_lanes:
        ahead = nread + 1
        loads = [readers.submit(load, c) for c, _ in todo[:ahead]]
        saves = []
        nxt = [0]

        warmed = threading.Event()

        def lane(k):
            p_k, stream = lanes[k]
            with torch.cuda.stream(stream):
                # buffers + code objects while the first cases are being read (0.5 s that the first case used to carry).  The lanes share the
                # lazily built weight images (UNetEngine.clone is shallow): lane 0 builds them all, the others start behind it -- on the host
                # (no image is composed twice) and on the device (their streams wait for the stream the images were built on)
                if k == 0:
                    try:
                        p_k.warm_up()
                    finally:
                        warmed.set()
                else:
                    warmed.wait()
                    stream.wait_stream(lanes[0][1])
                    p_k.warm_up()
                while True:
                    with lock:
                        i = nxt[0]
                        if i >= len(todo):
                            return
                        nxt[0] += 1
                        fut = loads[i]
                        loads[i] = None
                        if i + ahead < len(todo):
                            loads.append(readers.submit(load, todo[i + ahead][0]))
                    case, out = todo[i]
                    t0 = time.time()
                    vol, meta, t_load = fut.result()
                    t1 = time.time()
                    # crop, (resample,) z-score, sliding window, (resample,) argmax, un-crop: all on the device
                    data_d, props = preprocess_case_device(vol, meta['spacing'], plans, stage, 'cuda:0')
                    ta = time.time()
                    resampled = tuple(data_d.shape[1:]) != tuple(props['shape_after_crop'])
                    seg_d, prob_d = p_k.predict_device(data_d, return_prob=args.save_softmax or resampled)
                    tb = time.time()
                    held = []

                    def host_alloc(shape, dtype):
                        held.append(pool.get(shape, dtype))
                        return held[-1]
                    seg_full, prob0 = export_prediction_device(seg_d, prob_d, props, args.save_softmax,
                                                               host_alloc if use_pinned else None)         # synchronises this lane's stream
                    if isinstance(vol, torch.Tensor):
                        pool.put(vol)                                       # its H2D copy is behind the synchronisation above
                    t2 = time.time()
                    with tlock:
                        tsum['wait'] += t1 - t0; tsum['pre'] += ta - t1; tsum['predict'] += tb - ta; tsum['export'] += t2 - tb
                    with lock:
                        saves.append(writers.submit(save, out, case, seg_full, prob0, meta, tuple(held)))
                        t_case_done.append(time.time())
                    print('  %s: shape %s%s, read %.2f s (waited %.2f s), gpu %.2f s'
                          % (case, tuple(data_d.shape[1:]), ' (resampled)' if resampled else '', t_load, t1 - t0, t2 - t1), flush=True)

        for f in [gpu_lanes.submit(lane, k) for k in range(len(lanes))]:
            f.result()                                                       # re-raises a failed case
        for f in saves:
            f.result()                                                       # re-raises a failed write
    if todo:
        dt = time.time() - t_all
        steady = ''
        if len(todo) >= 4:          # rate once the pipeline is full: from the end of the first case's GPU work to the last file written
            steady = '; first case done after %.2f s, then %.2f cases/s' % (t_case_done[0] - t_all, (len(todo) - 1) / max(1e-9, time.time() - t_case_done[0]))
        print('  %d case(s) in %.2f s (%.2f cases/s%s)' % (len(todo), dt, len(todo) / dt, steady), flush=True)
        n = float(len(todo))
        print('  host time per case (ms; %d reader(s), %d lane(s), %d writer(s) run side by side): read %.0f | lane: wait for the read %.0f, '
              'preprocess + H2D %.0f, queue the forwards %.0f, export + D2H (synchronises) %.0f | write %.0f'
              % (nread, len(lanes), nwrite, 1e3 * tsum['read'] / n, 1e3 * tsum['wait'] / n, 1e3 * tsum['pre'] / n, 1e3 * tsum['predict'] / n,
                 1e3 * tsum['export'] / n, 1e3 * tsum['write'] / n), flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
