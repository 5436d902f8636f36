"""nnU-Net checkpoint / plans files in the directory layout DeepWMH installs and reads.

Layout (reference: /root/reference/deepwmh/pipeline/DCNN_multistage.py:608-631,
/root/reference/deepwmh/main/install_model.py:18-43,85-106, /root/reference/deepwmh/main/predict.py:104,139):

  <RESULTS_FOLDER>/nnUNet/3d_fullres/<Task>/nnUNetTrainerV2__nnUNetPlansv2.1/plans.pkl
  <RESULTS_FOLDER>/nnUNet/3d_fullres/<Task>/nnUNetTrainerV2__nnUNetPlansv2.1/all/model_best.model      (torch.save dict)
  <RESULTS_FOLDER>/nnUNet/3d_fullres/<Task>/nnUNetTrainerV2__nnUNetPlansv2.1/all/model_best.model.pkl  (pickle)

[nnU-Net, not in tree] `.model` = {'epoch','state_dict','optimizer_state_dict','lr_scheduler_state_dict',
'plot_stuff','best_stuff','amp_grad_scaler'}; `.model.pkl` = {'init','name','class','plans'} where `init` is
the positional argument tuple of nnUNetTrainer.__init__ (install_model.py:93-103 rewrites init[0], [2], [3]).
"""
import os
import pickle
from collections import OrderedDict

import numpy as np

from .plan import UNetPlan

TRAINER = 'nnUNetTrainerV2'
PLANS_ID = 'nnUNetPlansv2.1'
NETWORK = '3d_fullres'


def trainer_folder(results_folder, task, network=NETWORK, trainer=TRAINER, plans_id=PLANS_ID):
    return os.path.join(results_folder, 'nnUNet', network, task, '%s__%s' % (trainer, plans_id))


def resolve_task(results_folder, task, network=NETWORK):
    """`-t` may be a full name (TaskNNN_Name) or an integer id."""
    base = os.path.join(results_folder, 'nnUNet', network)
    if os.path.isdir(os.path.join(base, str(task))):
        return str(task)
    try:
        tid = int(task)
    except ValueError:
        raise FileNotFoundError('task "%s" not found under %s' % (task, base))
    hits = [d for d in sorted(os.listdir(base)) if d.startswith('Task%03d_' % tid)] if os.path.isdir(base) else []
    if len(hits) != 1:
        raise FileNotFoundError('task id %d: %d matching folders under %s' % (tid, len(hits), base))
    return hits[0]


def fold_folder(fold):
    return 'all' if str(fold) == 'all' else 'fold_%d' % int(fold)


def load_checkpoint(folder, fold='all', checkpoint_name='model_best'):
    """Returns (UNetPlan, plans dict, state_dict of numpy arrays, info dict)."""
    import torch
    fdir = os.path.join(folder, fold_folder(fold))
    model = os.path.join(fdir, checkpoint_name + '.model')
    pkl = model + '.pkl'
    for p in (model, pkl):
        if not os.path.isfile(p):
            raise FileNotFoundError('missing checkpoint file %s' % p)
    with open(pkl, 'rb') as f:
        info = pickle.load(f)
    plans = info['plans']
    stage = None
    init = info.get('init')
    if init is not None and len(init) > 5 and init[5] is not None:
        stage = init[5]
    plan = UNetPlan.from_nnunet_plans(plans, stage)
    ck = torch.load(model, map_location='cpu', weights_only=False)
    sd = OrderedDict()
    for k, v in ck['state_dict'].items():
        k = k[7:] if k.startswith('module.') else k
        sd[k] = v.detach().cpu().float().numpy()
    return plan, plans, sd, dict(epoch=ck.get('epoch'), init=init, name=info.get('name'), checkpoint=ck)


def make_plans(plan: UNetPlan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True, batch_size=2):
    """An nnU-Net-shaped plans dict for `plan` (what nnUNet_plan_and_preprocess would have pickled)."""
    st = OrderedDict(batch_size=batch_size, num_pool_per_axis=[plan.num_pool] * 3,
                     patch_size=np.array(plan.patch_size), median_patient_size_in_voxels=np.array(plan.patch_size),
                     current_spacing=np.array(spacing, dtype=np.float64), original_spacing=np.array(spacing, dtype=np.float64),
                     do_dummy_2D_data_aug=False, pool_op_kernel_sizes=[list(p) for p in plan.pool_op_kernel_sizes],
                     conv_kernel_sizes=[list(k) for k in plan.conv_kernel_sizes])
    nmod = plan.num_input_channels
    return OrderedDict(
        num_stages=1, num_modalities=nmod, modalities={i: 'FLAIR' for i in range(nmod)},
        normalization_schemes=OrderedDict((i, 'nonCT') for i in range(nmod)), dataset_properties={},
        list_of_npz_files=[], original_spacings=[], original_sizes=[], preprocessed_data_folder='',
        num_classes=plan.num_classes - 1, all_classes=list(range(1, plan.num_classes)),
        base_num_features=plan.base_num_features, use_mask_for_norm=OrderedDict((i, use_mask_for_norm) for i in range(nmod)),
        keep_only_largest_region=None, min_region_size_per_class=None, min_size_per_class=None,
        transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2], data_identifier='nnUNetData_plans_v2.1',
        plans_per_stage={0: st}, preprocessor_name='GenericPreprocessor', conv_per_stage=plan.conv_per_stage)


def save_checkpoint(folder, state_dict, plans, fold='all', checkpoint_name='model_best', epoch=0, extra=None):
    """Writes `<name>.model` + `<name>.model.pkl` (+ plans.pkl beside the fold folder) in nnU-Net's format."""
    import torch
    fdir = os.path.join(folder, fold_folder(fold))
    os.makedirs(fdir, exist_ok=True)
    sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
                     for k, v in state_dict.items())
    # no 'amp_grad_scaler' key: nnU-Net's load_checkpoint_ram only restores the scaler when the key exists, and a None
    # value would be passed to GradScaler.load_state_dict if real nnU-Net continued training from this file
    ck = dict(epoch=epoch, state_dict=sd, optimizer_state_dict=None, lr_scheduler_state_dict=None,
              plot_stuff=([], [], [], []), best_stuff=(None, None, None))
    if extra:
        ck.update(extra)
    model = os.path.join(fdir, checkpoint_name + '.model')
    plans_file = os.path.join(folder, 'plans.pkl')
    init = (plans_file, fold, folder, '', True, 0, True, False, True)

    def atomically(path, writer):
        # a kill in the middle of a write must not leave a truncated model_latest behind (the next `-c` would die on it)
        tmp = '%s.tmp%d' % (path, os.getpid())
        try:
            with open(tmp, 'wb') as f:
                writer(f)
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    atomically(model, lambda f: torch.save(ck, f))
    atomically(model + '.pkl', lambda f: pickle.dump(OrderedDict(
        init=init, name=TRAINER, **{'class': "<class 'nnunet.training.network_training.nnUNetTrainerV2.nnUNetTrainerV2'>"}, plans=plans), f))
    atomically(plans_file, lambda f: pickle.dump(plans, f))
    return model


class CheckpointProcess:
    """Checkpoint files written by a helper process (cli/_ckpt_writer.py: why a process and not a thread).  The trainer hands over raw
    snapshots of its flat parameter / momentum arenas (one file write each, no pickling on the training side); jobs are executed in order;
    `drain()` waits for everything submitted and raises if a write failed.  `plans` and the parameter layout are sent once."""

    def __init__(self, param_order, offsets, plans, scratch_dir):
        import json
        import subprocess
        import sys
        import tempfile
        self._json = json
        # snapshots are raw copies of the arenas (2 x 4 bytes per parameter and snapshot, two snapshots alive at most): memory-backed /dev/shm when it
        # has the room (containers often give it 64 MB), else next to the checkpoints
        need = 4 * 4 * int(offsets[-1] + int(np.prod(param_order[-1][1]))) + (64 << 20)
        shm = '/dev/shm'
        try:
            st = os.statvfs(shm)
            shm_ok = os.access(shm, os.W_OK) and st.f_bavail * st.f_frsize >= need
        except OSError:
            shm_ok = False
        self.dir = tempfile.mkdtemp(prefix='wmh_ckpt_', dir=shm if shm_ok else scratch_dir)
        plans_file = os.path.join(self.dir, 'plans.pkl')
        with open(plans_file, 'wb') as f:
            pickle.dump(plans, f)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='',
                   PYTHONPATH=root + (os.pathsep + os.environ['PYTHONPATH'] if os.environ.get('PYTHONPATH') else ''))
        self.proc = subprocess.Popen([sys.executable, '-m', 'deepwmh_amd.cli._ckpt_writer'], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     env=env, text=True, bufsize=1)
        self.pending = self._since_snap = 0
        self._send(dict(op='init', order=[[n, list(s)] for n, s in param_order], offsets=[int(v) for v in offsets], plans=plans_file))
        self._snap, self._n = None, 0

    def _send(self, job):
        self.proc.stdin.write(self._json.dumps(job) + '\n')
        self.proc.stdin.flush()
        self.pending += 1
        self._since_snap += 1

    def snapshot(self, params_flat, momentum_flat=None):
        """params_flat / momentum_flat: host float32 arrays (the whole arenas).  Files of the previous snapshot are dropped once the
        jobs that use them have run (the drop is queued behind them)."""
        # at most TWO snapshots alive (what `need` budgets): everything older than the current snapshot's own jobs -- in particular the
        # drop of the snapshot before it -- has to be acknowledged before a new one is written (a failed write surfaces here, too)
        self.drain(keep=self._since_snap)
        if self._snap:
            self._send(dict(op='drop', files=[f for f in self._snap if f]))
        self._since_snap = 0
        self._n += 1
        pf = os.path.join(self.dir, 'params_%d.f32' % self._n)
        params_flat.tofile(pf)
        mf = None
        if momentum_flat is not None:
            mf = os.path.join(self.dir, 'momentum_%d.f32' % self._n)
            momentum_flat.tofile(mf)
        self._snap = (pf, mf)

    def add_momentum(self, momentum_flat):
        pf, mf = self._snap
        if mf is None:
            mf = os.path.join(self.dir, 'momentum_%d.f32' % self._n)
            momentum_flat.tofile(mf)
            self._snap = (pf, mf)

    def save(self, folder, fold, name, epoch, extra, with_momentum=False):
        pf, mf = self._snap
        self._send(dict(op='save', params=pf, momentum=mf if with_momentum else None, folder=folder, fold=str(fold), name=name,
                        epoch=int(epoch), extra=extra))

    def drain(self, keep=0):
        while self.pending > keep:
            line = self.proc.stdout.readline()
            if not line:
                raise RuntimeError('checkpoint writer process ended unexpectedly (exit code %s)' % self.proc.poll())
            self.pending -= 1
            if not line.startswith('ok'):
                raise RuntimeError('checkpoint writer: ' + line.strip())

    def abort(self):
        """The training loop is going down with an exception: no waiting, no half-written promises -- end the helper, remove the arenas."""
        import shutil
        try:
            self.proc.kill()
            self.proc.wait(timeout=10)
        except Exception:
            pass
        shutil.rmtree(self.dir, ignore_errors=True)

    def close(self):
        import shutil
        try:
            self.drain()
        finally:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=60)
            except Exception:
                self.proc.kill()
            shutil.rmtree(self.dir, ignore_errors=True)
