"""Training data side of `nnUNet_plan_and_preprocess` / `nnUNet_train`: dataset fingerprint and plan, preprocessed
case storage, random patch sampling with foreground oversampling [nnU-Net, not in tree:
experiment_planning/experiment_planner_baseline_3DUNet_v21.py, preprocessing/, training/dataloading/dataset_loading.py
DataLoader3D].  Crop / resampling / normalisation of the cases run on the GPU (preprocess.py -> libdeepwmh_amd.so; no host
form in the product); file IO, the planner and the patch sampler are host code.
Directory contract: /root/reference/deepwmh/pipeline/DCNN_multistage.py:50-93 (raw), :620-631 (preprocessed)."""
import json
import os
import pickle
from collections import OrderedDict

import numpy as np

from . import nifti, preprocess
from .checkpoint import PLANS_ID, make_plans
from .plan import UNetPlan
from .planner import plan_3d_fullres

DATA_ID = 'nnUNetData_plans_v2.1'


def find_task(raw_base, task):
    base = os.path.join(raw_base, 'nnUNet_raw_data')
    try:
        tid = int(task)
        hits = [d for d in sorted(os.listdir(base)) if d.startswith('Task%03d_' % tid)]
    except ValueError:
        hits = [task] if os.path.isdir(os.path.join(base, str(task))) else []
    if len(hits) != 1:
        raise FileNotFoundError('task %s: %d matching folders under %s' % (task, len(hits), base))
    return hits[0]


def plan_and_preprocess(raw_base, preprocessed_base, task, verify=True, log=print):
    task = find_task(raw_base, task)
    raw = os.path.join(raw_base, 'nnUNet_raw_data', task)
    with open(os.path.join(raw, 'dataset.json')) as f:
        ds = json.load(f)
    nmod = len(ds['modality'])
    nclass = len(ds['labels'])
    cases = [os.path.basename(t['image'])[:-len('.nii.gz')] for t in ds['training']]
    out = os.path.join(preprocessed_base, task)
    stage_dir = os.path.join(out, DATA_ID + '_stage0')
    os.makedirs(stage_dir, exist_ok=True)
    loaded, spacings, shapes = [], [], []
    for c in cases:
        imgs = []
        meta = None
        for m in range(nmod):
            p = os.path.join(raw, 'imagesTr', '%s_%04d.nii.gz' % (c, m))
            if not os.path.isfile(p):
                raise FileNotFoundError('dataset integrity: missing %s' % p)
            arr, mt = nifti.read(p)
            imgs.append(arr)
            meta = meta or mt
        lab, lmeta = nifti.read(os.path.join(raw, 'labelsTr', c + '.nii.gz'))
        if verify:
            if lab.shape != imgs[0].shape:
                raise ValueError('dataset integrity: %s image %s vs label %s' % (c, imgs[0].shape, lab.shape))
            if not np.allclose(lmeta['spacing'], meta['spacing'], atol=1e-3):
                raise ValueError('dataset integrity: %s spacing differs between image and label' % c)
            u = np.unique(lab)
            if not set(u.astype(int).tolist()) <= set(range(nclass)):
                raise ValueError('dataset integrity: %s has labels %s outside dataset.json' % (c, u))
        data = np.stack(imgs).astype(np.float32)
        cropped, seg, nz, bbox = preprocess.crop_training_case_device(data, lab)
        loaded.append((c, cropped, seg, nz, bbox, meta, data.shape[1:]))
        spacings.append(meta['spacing'])
        shapes.append(cropped.shape[1:])
    # ExperimentPlanner3D_v21 (deepwmh_amd/planner.py): target spacing, axis order, pooling / kernel sizes per stage, patch, batch
    target_spacing, tf, tb, stage = plan_3d_fullres(spacings, shapes, nmod, nclass, log)
    patch, num_pool = [int(v) for v in stage['patch_size']], len(stage['pool_op_kernel_sizes'])
    median_shape = stage['median_patient_size_in_voxels']
    crop_frac = np.median([np.prod(sh) / np.prod(full) for (_, _, _, _, _, _, full), sh in zip(loaded, shapes)])
    plan = UNetPlan(nmod, 32, nclass, stage['pool_op_kernel_sizes'], stage['conv_kernel_sizes'], 2, 320, patch)
    plan.check_supported()          # before any case is preprocessed or written: an unrunnable plan is a failed command
    plans = make_plans(plan, tuple(float(v) for v in stage['current_spacing']), use_mask_for_norm=bool(crop_frac < 0.75),
                       batch_size=int(stage['batch_size']))
    plans['transpose_forward'], plans['transpose_backward'] = list(tf), list(tb)
    plans['original_spacings'] = [np.array(s) for s in spacings]
    plans['original_sizes'] = [np.array(s) for s in shapes]
    plans['preprocessed_data_folder'] = out
    st0 = plans['plans_per_stage'][0]
    st0['median_patient_size_in_voxels'] = median_shape
    st0['num_pool_per_axis'] = list(stage['num_pool_per_axis'])
    st0['do_dummy_2D_data_aug'] = stage['do_dummy_2D_data_aug']
    use_mask = plans['use_mask_for_norm'][0]
    target_t = np.asarray(stage['current_spacing'], float)
    for (c, cropped, seg, nz, bbox, meta, full) in loaded:
        # network axis order (transpose_forward), then resampling (the low-resolution axis on its own when anisotropic) as in
        # GenericPreprocessor.resample_and_normalize; segmentation label-wise linear
        cropped_t = cropped.transpose([0] + [i + 1 for i in tf])
        seg_t, nz_t = seg.transpose(tf), nz.transpose(tf)
        sp_t = np.asarray(meta['spacing'], float)[tf]
        norm, seg_r = preprocess.resample_normalize_training_case_device(cropped_t, seg_t, nz_t, sp_t, target_t, use_mask)
        res = norm
        allc = np.concatenate([norm, seg_r[None].astype(np.float32)], 0)
        np.savez_compressed(os.path.join(stage_dir, c + '.npz'), data=allc)
        fg = np.argwhere(seg_r > 0)
        if len(fg) > 10000:
            fg = fg[np.random.default_rng(1234).choice(len(fg), 10000, replace=False)]
        with open(os.path.join(stage_dir, c + '.pkl'), 'wb') as f:
            pickle.dump(OrderedDict(original_size_of_raw_data=np.array(full), original_spacing=np.array(meta['spacing']),
                                    crop_bbox=bbox, size_after_cropping=cropped.shape[1:], size_after_resampling=res.shape[1:],
                                    spacing_after_resampling=target_t, class_locations={1: fg}), f)
    for name in ('%s_plans_3D.pkl' % PLANS_ID, '%s_plans_2D.pkl' % PLANS_ID):
        with open(os.path.join(out, name), 'wb') as f:
            pickle.dump(plans, f)
    with open(os.path.join(out, 'dataset_properties.pkl'), 'wb') as f:
        pickle.dump(OrderedDict(all_sizes=shapes, all_spacings=spacings, all_classes=list(range(1, nclass)),
                                modalities=ds['modality'], size_reductions=OrderedDict()), f)
    with open(os.path.join(out, 'dataset.json'), 'w') as f:
        json.dump(ds, f, indent=4)
    log('planned %s: %d cases, axis order %s, median shape %s, target spacing %s, patch %s, batch %d, poolings %s, kernels %s, '
        'use_mask_for_norm %s' % (task, len(cases), tf, np.asarray(median_shape).tolist(), np.round(target_t, 3).tolist(), patch,
                                  int(stage['batch_size']), stage['pool_op_kernel_sizes'], stage['conv_kernel_sizes'], use_mask))
    return task, plans


class PatchLoader:
    """DataLoader3D: random patches, a third of each batch forced onto foreground, zero-padding when a case is
    smaller than the patch; mirroring along random axes (the one augmentation done here; nnU-Net's rotation /
    scaling / gamma / noise transforms are host-side batchgenerators code and are not reproduced)."""

    def __init__(self, stage_dir, cases, patch, batch_size=2, oversample_fg=0.33, seed=0, mirror=True):
        self.patch, self.bs, self.ofg, self.mirror = tuple(patch), batch_size, oversample_fg, mirror
        self.rng = np.random.default_rng(seed)
        self.cases = []
        for c in cases:
            d = np.load(os.path.join(stage_dir, c + '.npz'))['data']
            with open(os.path.join(stage_dir, c + '.pkl'), 'rb') as f:
                props = pickle.load(f)
            self.cases.append((c, d, props.get('class_locations', {}).get(1, np.zeros((0, 3), int))))
        if not self.cases:
            raise ValueError('no training cases')

    def next_batch(self):
        P = self.patch
        data = np.zeros((self.bs, self.cases[0][1].shape[0] - 1) + P, np.float32)
        seg = np.zeros((self.bs,) + P, np.uint8)
        for b in range(self.bs):
            _, d, fg = self.cases[int(self.rng.integers(len(self.cases)))]
            shp = d.shape[1:]
            force_fg = b >= round(self.bs * (1 - self.ofg)) and len(fg) > 0
            lo = []
            if force_fg:
                ctr = fg[int(self.rng.integers(len(fg)))]
            for ax in range(3):
                room = shp[ax] - P[ax]
                if force_fg:
                    start = int(np.clip(ctr[ax] - P[ax] // 2, min(0, room), max(0, room)))
                else:
                    start = int(self.rng.integers(min(0, room), max(0, room) + 1)) if room != 0 else 0
                lo.append(start)
            src = tuple(slice(max(0, l), min(s, l + p)) for l, s, p in zip(lo, shp, P))
            dst = tuple(slice(max(0, -l), max(0, -l) + (sl.stop - sl.start)) for l, sl in zip(lo, src))
            data[(b, slice(None)) + dst] = d[(slice(None, -1),) + src]
            s = d[(-1,) + src]
            seg[(b,) + dst] = np.maximum(s, 0).astype(np.uint8)
            if self.mirror:
                for ax in range(3):
                    if self.rng.random() < 0.5:
                        data[b] = np.flip(data[b], ax + 1)
                        seg[b] = np.flip(seg[b], ax)
        return np.ascontiguousarray(data), np.ascontiguousarray(seg)


class PrefetchLoader:
    """Runs a PatchLoader in a background thread (nnU-Net's MultiThreadedAugmenter does this with worker processes):
    batches are produced `depth` ahead so that host-side sampling overlaps the GPU step.  Order and content are exactly
    the wrapped loader's (one producer, FIFO queue), so seeded runs stay reproducible."""

    def __init__(self, loader, depth=3):
        import queue
        import threading
        self.loader = loader
        self.q = queue.Queue(maxsize=depth)
        self._stop = False
        self._err = None

        def work():
            try:
                while not self._stop:
                    item = loader.next_batch()
                    while not self._stop:
                        try:
                            self.q.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            continue
            except BaseException as e:           # surfaced on the consumer side
                self._err = e
                self.q.put(None)

        self.t = threading.Thread(target=work, daemon=True)
        self.t.start()

    def next_batch(self):
        item = self.q.get()
        if item is None:
            raise RuntimeError('patch loader thread failed') from self._err
        return item

    def close(self):
        self._stop = True
