"""The nnU-Net training step on the HIP kernels: forward with deep supervision, Dice+CE loss, backward,
gradient clipping, SGD with Nesterov momentum.

Host-side mirror of `nnUNetTrainerV2.run_iteration` / `initialize_optimizer_and_scheduler` / `maybe_update_lr`
[nnU-Net, not in tree], i.e. what `nnUNet_train 3d_fullres nnUNetTrainerV2 <id> all ...` runs
(/root/reference/deepwmh/pipeline/DCNN_multistage.py:292-309,501-520), in fp32 (the reference runs AMP fp16).
Parameters, gradients and momentum live in three flat HBM arenas (one kernel launch each for clipping and
the optimizer; one RCCL all-reduce for data parallelism).  Nothing here falls back to torch autograd.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .network import IN_EPS, LRELU_SLOPE
from .plan import UNetPlan


def deep_supervision_weights(num_pool):
    """nnUNetTrainerV2.initialize: 1/2^i per scale, the lowest resolution masked out, normalised."""
    w = np.array([1 / (2 ** i) for i in range(num_pool)], dtype=np.float64)
    if num_pool > 1:
        w[-1] = 0
    return w / w.sum()


def poly_lr(epoch, max_epochs, initial_lr=1e-2, exponent=0.9):
    return initial_lr * (1 - epoch / max_epochs) ** exponent


class _Conv:
    """kernel (kz,3,3), stride (sz, stride, stride): kz = 1 / sz = 1 on nnU-Net's anisotropic stages (thick-slice data)."""

    def __init__(self, name, cin, cout, stride, kz=3, sz=0):
        self.name, self.cin, self.cout, self.stride = name, cin, cout, stride
        self.kz, self.sz = kz, (sz or stride)
        self.iso = kz == 3 and self.sz == stride


class UNetTrainer:
    def __init__(self, plan: UNetPlan, state_dict, device='cuda:0', momentum=0.99, weight_decay=3e-5, clip=12.0, precision='fp32'):
        """precision 'fp32' (the exact-fp32 kernels) or 'f16x2'.  (Rounds 1-5 also had a 'bf16' mixed-precision mode -- bf16 matrix-core
        operands in the wide layers; gradients 8.6 % off float64 autograd on the tiny net -- removed in round 6: f16x2 is as fast and exact.)"""
        if precision not in ('fp32', 'f16x2'):
            raise ValueError('precision must be fp32 or f16x2')
        self.precision = precision
        # 'f16x2': forward and stride-1 data-gradient convolutions with rows of >= 16 voxels form their products from fp16 pairs
        # (network.py, DESIGN.md 3.2: fp32-grade, 2-2.7x the fp32 instruction's speed); weight gradients and everything else fp32
        self.x3 = precision == 'f16x2'
        self.x3_s2_dgrad = self.x3_tu_dgrad = True         # diagnostic attributes (tools/grad_compare.py): False = those data gradients on the exact-fp32 kernels
        self.x3_s2_wgrad = True                            # diagnostic attribute: False = the strided weight gradients on the exact-fp32 kernel
        self.x3_splitk = True                              # diagnostic attribute (tools/train_bench.py): False = no slab memory for the pair kernels (no split-K)
        # fp16's exponent range: the data gradients (dLoss/dy ~ 1e-8 .. 1e-2 at 2 x 128^3 voxels) sit below fp16's normal range, so in
        # this mode the backward pass runs on the loss scaled by 2^16 -- the static form of the GradScaler nnU-Net's own fp16 AMP
        # training uses -- and loss and gradient arena are scaled back (exactly: a power of two) at the end of backward().  Without it
        # the full-size gradients were 7 % off (tools/grad_compare.py); the forward pass (O(1) activations) needs nothing.
        # The scale is dynamic like GradScaler's (init 2^16, x0.5 after a step whose gradient norm was not finite, x2 after 2000 clean
        # steps, never above 2^16 -- the value the full-size gradients were validated at): the device skips such a step on its own
        # (wmh_grad_clip_coef reports the norm, wmh_sgd_nesterov_step leaves parameters and momentum alone), the host learns of it one
        # iteration later from a pinned copy of the norm (no synchronisation on the step's own stream work).
        self.loss_scale = 65536.0 if self.x3 else 1.0
        self.max_loss_scale, self.min_loss_scale, self.growth_interval = self.loss_scale, 1.0, 2000
        self.skipped_steps, self._clean_steps, self._norm_event = 0, 0, None
        plan.check_supported()
        self.plan, self.device = plan, torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('UNetTrainer needs a HIP device; there is no CPU path')
        self.lib = _lib.load()
        _lib.check(self.lib.wmh_device_check(self.device.index or 0), 'wmh_device_check')
        self.momentum, self.weight_decay, self.clip = momentum, weight_decay, clip
        self.K = plan.num_classes
        self.ds_weights = deep_supervision_weights(plan.num_pool)
        self.enc = [_Conv(s.name, s.cin, s.cout, s.stride, s.kz, s.sz) for s in plan.encoder_convs()]
        for c in self.enc:
            if c.kz == 1 and c.sz == 2:
                raise NotImplementedError('%s: kernel 1 with stride 2 along the same axis has no kernel' % c.name)
        self.dec = []
        for st in plan.decoder_stages():
            name, cdown, cskip = st['tu']
            self.dec.append(dict(tu=name, cdown=cdown, cskip=cskip, seg=st['seg'], kt=int(st['tu_kt']),
                                 convs=[_Conv(c.name, c.cin, c.cout, 1, c.kz, 1) for c in st['convs']]))
        # ---- flat arenas: upstream parameter order (localization, context, tu, seg_outputs) ------------------
        order = []
        for st in self.dec:
            for c in st['convs']:
                order += [(c.name + '.conv.weight', (c.cout, c.cin, c.kz, 3, 3)), (c.name + '.conv.bias', (c.cout,)),
                          (c.name + '.instnorm.weight', (c.cout,)), (c.name + '.instnorm.bias', (c.cout,))]
        for c in self.enc:
            order += [(c.name + '.conv.weight', (c.cout, c.cin, c.kz, 3, 3)), (c.name + '.conv.bias', (c.cout,)),
                      (c.name + '.instnorm.weight', (c.cout,)), (c.name + '.instnorm.bias', (c.cout,))]
        for st in self.dec:
            order.append((st['tu'] + '.weight', (st['cdown'], st['cskip'], st['kt'], 2, 2)))
        for st in self.dec:
            order.append((st['seg'] + '.weight', (self.K, st['cskip'], 1, 1, 1)))
        self.param_order = order
        sizes = [int(np.prod(s)) for _, s in order]
        offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])])       # 16-byte aligned views
        self.nparams = int(offs[-1])
        dev = self.device
        self.params = torch.zeros(self.nparams, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.nparams, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(self.nparams, dtype=torch.float32, device=dev)
        self.p, self.g = {}, {}
        for (name, shape), o, n in zip(order, offs[:-1], sizes):
            self.p[name] = self.params[int(o):int(o) + n].view(*shape)
            self.g[name] = self.grads[int(o):int(o) + n].view(*shape)
        # element range of every parameter in the arena, alignment padding included (data-parallel exchange, run_iteration)
        self.grad_ranges = {name: (int(offs[i]), int(offs[i + 1])) for i, (name, _) in enumerate(order)}
        self._avg = None
        self.load_state_dict(state_dict)
        self.first_step = True
        # weight gradients on a second stream beside the InstanceNorm backward of the next layer (WMH_TRAIN_OVERLAP=0: one stream)
        self.overlap_wgrad = os.environ.get('WMH_TRAIN_OVERLAP', '1') != '0'
        self._side, self._wgrad_pending, self._side_retired = None, [], []
        self._bufs, self._packed, self._version = {}, {}, 0
        self._x3_jobs, self._x3_table, self._x3_batched_version, self._x3_used = OrderedDict(), None, -1, {}
        self.clip_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self._norm_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        self.loss_acc = torch.zeros(1, dtype=torch.float64, device=dev)
        self.ws_d = torch.empty(max(self.lib.wmh_dice_ce_loss_workspace_doubles(), 1024), dtype=torch.float64, device=dev)

    # ---- parameters --------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        for name, shape in self.param_order:
            v = sd[name]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
            self.p[name].copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
        self._version = getattr(self, '_version', 0) + 1

    def state_dict(self):
        return OrderedDict((name, self.p[name].detach().cpu().numpy().copy()) for name, _ in self.param_order)

    def momentum_state(self):
        return self.mom.detach().cpu().numpy().copy()

    # ---- plumbing -----------------------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype=torch.float32):
        n = int(np.prod(shape))
        t = self._bufs.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            if t is not None and key.startswith('b.wgrad.ws.side'):
                # a workspace of the weight-gradient stream that has to grow: the kernel that used the old one may still be running there, and
                # the caching allocator only knows the stream it was allocated on -- keep it alive until the streams have joined (backward())
                self._side_retired.append(t)
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t[:n].view(*shape)

    @staticmethod
    def _st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _pack(self, conv, cfg, mode, ci_off, ci_cnt):
        """Fragment-major weight image for the current parameters (repacked on the device once per step)."""
        key = (conv.name, cfg, mode, ci_off, ci_cnt)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        ecin, ecout = (ci_cnt, conv.cout) if mode == 0 else (conv.cout, ci_cnt)
        n = self.lib.wmh_conv3d_k3_packed_floats(cfg, ecin, ecout)
        wf = hit[1] if hit is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wmh_conv3d_k3_pack_weights_device(cfg, _lib.ptr(self.p[conv.name + '.conv.weight']), conv.cin,
                                                              conv.cout, ci_off, ci_cnt, mode, _lib.ptr(wf), self._st()),
                   'wmh_conv3d_k3_pack_weights_device')
        self._packed[key] = (self._version, wf)
        return wf

    def _use_x3(self, cfg_b, stride, C0, C1, W, fwd=False):
        # the forward pass also has pair kernels for the strided layers (configurations 1, 3, 5: as the inference engine uses them);
        # their data gradients stay on the fp32 gather kernel
        if stride == 2 and not (fwd and cfg_b in (1, 3, 5, 25, 26)):
            return False
        return (self.x3 and (C1 == 0 or C0 % 8 == 0) and (W % 4 == 0 or cfg_b in (3, 4, 5)) and bool(self.lib.wmh_conv3d_k3_f16x2_supported(cfg_b)))

    def _pack_x3(self, conv, cfg, mode, ci_off, ci_cnt):
        key = (conv.name, cfg, mode, ci_off, ci_cnt, 'x3')
        hit = self._packed.get(key)
        self._x3_used[key] = self._version                   # (repack_x3_batch drops the images nobody asked for during the last step)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        ecin, ecout = (ci_cnt, conv.cout) if mode == 0 else (conv.cout, ci_cnt)
        n = self.lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, ecin, ecout)
        wf = hit[1] if hit is not None else torch.empty(n, dtype=torch.int16, device=self.device)
        args = (cfg, self.p[conv.name + '.conv.weight'], conv.cin, conv.cout, ci_off, ci_cnt, mode, wf)
        self._pack_x3_launch(key, args)
        return wf

    def _pack_x3_launch(self, key, args):
        """One image build now; the build is remembered so that from the next step on all of a step's images are rebuilt by ONE launch
        (repack_x3_batch).  args = (cfg, weight tensor, Cin_total, Cout, ci_off, ci_cnt, mode, image tensor)."""
        cfg, w, cin, cout, ci_off, ci_cnt, mode, wf = args
        _lib.check(self.lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, _lib.ptr(w), cin, cout, ci_off, ci_cnt, mode, _lib.ptr(wf), self._st()),
                   'wmh_conv3d_k3_pack_weights_f16x2_device')
        self._packed[key] = (self._version, wf)
        job = self._x3_jobs.get(key)
        if job is None or job[7] is not wf or job[1] is not w:      # a new image, or one whose buffers were replaced
            self._x3_jobs[key] = args
            self._x3_table = None                            # the device table is rebuilt before its next use

    def repack_x3_batch(self):
        """All f16x2 weight images this trainer has built so far, rebuilt from the current parameters in one launch (the parameters and the
        image buffers live at fixed addresses, so the job table is uploaded once).  A training step rebuilt 43 images one launch each:
        0.6 ms of a 31 ms iteration.  Bit-identical to the single builds (tests/test_gpu_trainer.py)."""
        # images not requested while the previous parameter version was current (another batch / patch shape, a diagnostic switch
        # flipped) are neither rebuilt nor kept alive by their job
        stale = [k for k in self._x3_jobs if self._x3_used.get(k, -1) < self._version - 1]
        for k in stale:
            del self._x3_jobs[k]
            self._x3_used.pop(k, None)
            self._packed.pop(k, None)
        if stale:
            self._x3_table = None
        if not self._x3_jobs:
            return
        if self._x3_table is None:
            nb = int(self.lib.wmh_conv3d_k3_pack_job_bytes())
            host = np.zeros((len(self._x3_jobs), nb), dtype=np.uint8)
            for i, (cfg, w, cin, cout, ci_off, ci_cnt, mode, wf) in enumerate(self._x3_jobs.values()):
                _lib.check(self.lib.wmh_conv3d_k3_pack_job_f16x2(cfg, _lib.ptr(w), cin, cout, ci_off, ci_cnt, mode, _lib.ptr(wf),
                                                                 C.c_void_p(host[i].ctypes.data)), 'wmh_conv3d_k3_pack_job_f16x2')
            self._x3_table = torch.from_numpy(host).to(self.device)
        _lib.check(self.lib.wmh_conv3d_k3_pack_weights_f16x2_batch(_lib.ptr(self._x3_table), len(self._x3_jobs), self._st()),
                   'wmh_conv3d_k3_pack_weights_f16x2_batch')
        for key, args in self._x3_jobs.items():
            self._packed[key] = (self._version, args[7])

    def _run_conv_x3(self, key, cfg, x0, ss0, x1, ss1, wf, bias, cout, want_stats, stride=1):
        N, C0, D, H, W = x0.shape
        C1 = 0 if x1 is None else x1.shape[1]
        Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
        y = self._buf(key + '.y', (N, cout, Do, Ho, Wo))
        part, npart = None, 0
        if want_stats:
            npart = _lib.check(self.lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, Do, Ho, Wo), 'num_partials_f16x2')
            part = self._buf(key + '.part', (N, cout, npart, 4))
        nws = self.lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, C0 + C1, N, D, H, W, cout) if self.x3_splitk else 0
        ws = self._buf('splitk.ws', (nws,)) if nws else None
        _lib.check(self.lib.wmh_conv3d_k3_fwd_f16x2(cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wf),
                                                     _lib.ptr(bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws), N, D, H, W, cout, LRELU_SLOPE,
                                                     self._st()),
                   'wmh_conv3d_k3_fwd_f16x2')
        return y, part, npart

    def _run_conv(self, key, cfg, x0, ss0, x1, ss1, wf, bias, cout, stride, want_stats, sz=0):
        N, C0, D, H, W = x0.shape
        C1 = 0 if x1 is None else x1.shape[1]
        Do, Ho, Wo = (D - 1) // (sz or stride) + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
        y = self._buf(key + '.y', (N, cout, Do, Ho, Wo))
        part, npart = None, 0
        if want_stats:
            npart = _lib.check(self.lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo), 'num_partials')
            part = self._buf(key + '.part', (N, cout, npart, 4))
        nws = self.lib.wmh_conv3d_k3_workspace_floats(cfg, C0 + C1, N, D, H, W, cout, stride)
        ws = self._buf('splitk.ws', (nws,)) if nws else None
        _lib.check(self.lib.wmh_conv3d_k3_fwd(cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wf),
                                              _lib.ptr(bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws), N, D, H, W, cout, stride,
                                              LRELU_SLOPE, self._st()), 'wmh_conv3d_k3_fwd')
        return y, part, npart

    # ---- forward -------------------------------------------------------------------------------------------------
    def _conv_fwd(self, key, conv, x0, ss0, x1=None, ss1=None):
        N, _, D, H, W = x0.shape
        cfg = _lib.check(self.lib.wmh_conv3d_config(conv.cin, conv.cout, D, H, W, conv.kz, conv.sz, conv.stride), 'config')
        C1 = 0 if x1 is None else x1.shape[1]
        cfg_x3 = self.lib.wmh_conv3d_k3_f16x2_config(cfg, conv.cout, W) if (self.x3 and conv.iso) else -1
        if cfg_x3 >= 0 and self._use_x3(cfg_x3, conv.stride, x0.shape[1], C1, W, fwd=True):
            wf = self._pack_x3(conv, cfg_x3, 0, 0, conv.cin)
            y, part, npart = self._run_conv_x3(key, cfg_x3, x0, ss0, x1, ss1, wf, self.p[conv.name + '.conv.bias'], conv.cout, True, conv.stride)
        else:
            wf = self._pack(conv, cfg, 0, 0, conv.cin)
            y, part, npart = self._run_conv(key, cfg, x0, ss0, x1, ss1, wf, self.p[conv.name + '.conv.bias'], conv.cout,
                                            conv.stride, True, conv.sz)
        ss = self._buf(key + '.ss', (N, conv.cout, 2))
        mr = self._buf(key + '.mr', (N, conv.cout, 2))
        _lib.check(self.lib.wmh_instnorm_finalize(_lib.ptr(part), npart, _lib.ptr(self.p[conv.name + '.instnorm.weight']),
                                                  _lib.ptr(self.p[conv.name + '.instnorm.bias']), IN_EPS, _lib.ptr(ss),
                                                  _lib.ptr(mr), N, conv.cout, self._st()), 'instnorm_finalize')
        return dict(key=key, conv=conv, x0=x0, ss0=ss0, x1=x1, ss1=ss1, y=y, ss=ss, mr=mr)

    def forward(self, x):
        """x [N,Cin,P0,P1,P2] on the device.  Returns the deep-supervision logits (full resolution first) and keeps
        the tape for backward()."""
        assert x.is_contiguous() and x.dtype == torch.float32
        if self.x3 and self._x3_jobs and self._x3_batched_version != self._version:
            self.repack_x3_batch()                           # every image the previous iterations used, in one launch
            self._x3_batched_version = self._version
        tape = dict(enc=[], dec=[])
        cur, cur_ss, skips = x, None, []
        nenc = len(self.enc)
        for i, conv in enumerate(self.enc):
            rec = self._conv_fwd('t.enc%d' % i, conv, cur, cur_ss)
            tape['enc'].append(rec)
            cur, cur_ss = rec['y'], rec['ss']
            if i % 2 == 1 and i < nenc - 2:
                skips.append((cur, cur_ss))
        logits = []
        for u, st in enumerate(self.dec):
            N, Cin, D, H, W = cur.shape
            kt = st['kt']
            up = self._buf('t.tu%d.y' % u, (N, st['cskip'], kt * D, 2 * H, 2 * W))
            wtu = self.p[st['tu'] + '.weight']
            if self.x3:                                      # pair form (any voxel count), image re-split on the device each step
                wf = self._pack_tu_x3(u, wtu, st)
                _lib.check(self.lib.wmh_convtranspose3d_fwd_f16x2(_lib.ptr(cur), _lib.ptr(cur_ss), _lib.ptr(wf), _lib.ptr(up), N,
                                                                  Cin, st['cskip'], D, H, W, kt, LRELU_SLOPE, self._st()), 'convT fwd f16x2')
            elif (D * H * W) % 4 == 0:
                wf = self._pack_tu(u, wtu, st)
                _lib.check(self.lib.wmh_convtranspose3d_mfma_fwd(_lib.ptr(cur), _lib.ptr(cur_ss), _lib.ptr(wf), _lib.ptr(up), N,
                                                                 Cin, st['cskip'], D, H, W, kt, LRELU_SLOPE, self._st()), 'convT fwd')
            elif kt != 2:
                raise NotImplementedError('(1,2,2) transposed convolution needs D*H*W % 4 == 0')
            else:
                _lib.check(self.lib.wmh_convtranspose3d_k2s2_fwd(_lib.ptr(cur), _lib.ptr(cur_ss), _lib.ptr(wtu), _lib.ptr(up), N, Cin,
                                                                 st['cskip'], D, H, W, LRELU_SLOPE, self._st()), 'convT fwd')
            skip, skip_ss = skips[-(u + 1)]
            r0 = self._conv_fwd('t.dec%d.0' % u, st['convs'][0], up, None, skip, skip_ss)
            r1 = self._conv_fwd('t.dec%d.1' % u, st['convs'][1], r0['y'], r0['ss'])
            lg = self._buf('t.logits%d' % u, (N, self.K) + tuple(r1['y'].shape[2:]))
            sp = int(np.prod(r1['y'].shape[2:]))
            _lib.check(self.lib.wmh_seghead_logits(_lib.ptr(r1['y']), _lib.ptr(r1['ss']), _lib.ptr(self.p[st['seg'] + '.weight']),
                                                   self.K, _lib.ptr(lg), N, st['cskip'], sp, LRELU_SLOPE, self._st()), 'seghead')
            tape['dec'].append(dict(tu_in=(cur, cur_ss), up=up, r0=r0, r1=r1, logits=lg))
            logits.append(lg)
            cur, cur_ss = r1['y'], r1['ss']
        self.tape = tape
        return logits[::-1]

    def _pack_tu(self, u, wtu, st):
        key = ('tu', u)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        n = self.lib.wmh_convtranspose3d_packed_floats(st['cdown'], st['cskip'], st['kt'])
        wf = hit[1] if hit is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wmh_convtranspose3d_pack_weights_device(_lib.ptr(wtu), st['cdown'], st['cskip'], st['kt'], _lib.ptr(wf), self._st()),
                   'wmh_convtranspose3d_pack_weights_device')
        self._packed[key] = (self._version, wf)
        return wf

    def _pack_tu_x3(self, u, wtu, st):
        key = ('tu.x3', u)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        n = self.lib.wmh_convtranspose3d_packed_dwords_f16x2(st['cdown'], st['cskip'], st['kt'])
        wf = hit[1] if hit is not None else torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.wmh_convtranspose3d_pack_weights_f16x2_device(_lib.ptr(wtu), st['cdown'], st['cskip'], st['kt'], _lib.ptr(wf), self._st()),
                   'wmh_convtranspose3d_pack_weights_f16x2_device')
        self._packed[key] = (self._version, wf)
        return wf

    def _pack_tu_dgrad(self, u, wtu, cfg, cin, cout):
        """Image of the transposed convolution's data gradient as a stride-2 convolution cout -> cin over dY (pack mode 2)."""
        key = ('tu.dgrad', u, cfg)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self._version:
            return hit[1]
        n = self.lib.wmh_conv3d_k3_packed_floats(cfg, cout, cin)
        wf = hit[1] if hit is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.wmh_conv3d_k3_pack_weights_device(cfg, _lib.ptr(wtu), cout, cin, 0, cout, 2, _lib.ptr(wf), self._st()),
                   'wmh_conv3d_k3_pack_weights_device')
        self._packed[key] = (self._version, wf)
        return wf

    # ---- backward ------------------------------------------------------------------------------------------------
    def _block_bwd(self, rec, g1, g2, dl, wseg_name, need_dx):
        """Backward of one Conv-IN-LReLU block given the upstream gradient of its activated output
        (g1 + g2 + head).  Returns the gradient(s) of the block's activated input(s)."""
        lib, st = self.lib, self._st()
        conv, y = rec['conv'], rec['y']
        N, Cc = y.shape[:2]
        S = int(np.prod(y.shape[2:]))
        dy = self._buf('b.dy.%s' % rec['key'], y.shape)
        sums = self._buf('b.sums', (lib.wmh_in_lrelu_bwd_scratch_doubles(N, Cc, S),), torch.float64)
        _lib.check(lib.wmh_in_lrelu_bwd(_lib.ptr(y), _lib.ptr(rec['ss']), _lib.ptr(rec['mr']), _lib.ptr(self.p[conv.name + '.instnorm.weight']),
                                        _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(dl), _lib.ptr(self.p[wseg_name]) if wseg_name else None,
                                        self.K, _lib.ptr(dy), _lib.ptr(self.g[conv.name + '.instnorm.weight']),
                                        _lib.ptr(self.g[conv.name + '.instnorm.bias']),
                                        _lib.ptr(self.g[wseg_name]) if (wseg_name and dl is not None) else None,
                                        _lib.ptr(self.g[conv.name + '.conv.bias']), _lib.ptr(sums),
                                        N, Cc, S, LRELU_SLOPE, st), 'wmh_in_lrelu_bwd')
        x0, x1 = rec['x0'], rec['x1']
        _, C0, D, H, W = x0.shape
        C1 = 0 if x1 is None else x1.shape[1]
        wname = conv.name + '.conv.weight'
        rest = [conv.name + '.conv.bias', conv.name + '.instnorm.weight', conv.name + '.instnorm.bias'] + ([wseg_name] if (wseg_name and dl is not None) else [])

        def wgrad(st, wskey):
            if not conv.iso:                                     # anisotropic stage: the general fp32 kernel
                nws = lib.wmh_conv3d_wgrad_workspace_floats(C0 + C1, Cc, N, D, H, W, conv.kz, conv.sz, conv.stride)
                ws = self._buf(wskey, (nws,))
                _lib.check(lib.wmh_conv3d_wgrad(_lib.ptr(x0), _lib.ptr(rec['ss0']), C0, _lib.ptr(x1), _lib.ptr(rec['ss1']), C1, _lib.ptr(dy),
                                                _lib.ptr(self.g[conv.name + '.conv.weight']), _lib.ptr(ws), N, D, H, W, Cc, conv.kz, conv.sz,
                                                conv.stride, LRELU_SLOPE, st), 'wmh_conv3d_wgrad')
            elif self.x3 and (conv.stride == 1 or self.x3_s2_wgrad) and lib.wmh_conv3d_k3_wgrad_f16x2_supported(C0 + C1, Cc, N, D, H, W, conv.stride):
                # pair form (fp16 hi + lo of dy and of the activated input): the loss scale keeps dy inside fp16's range
                s2 = conv.stride == 2                            # (round 5: the strided layers with output rows of >= 12 voxels too)
                nws = (lib.wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats if s2 else lib.wmh_conv3d_k3_wgrad_f16x2_workspace_floats)(C0 + C1, Cc, N, D, H, W)
                ws = self._buf(wskey, (nws,))
                fn = lib.wmh_conv3d_k3s2_wgrad_f16x2 if s2 else lib.wmh_conv3d_k3_wgrad_f16x2
                _lib.check(fn(_lib.ptr(x0), _lib.ptr(rec['ss0']), C0, _lib.ptr(x1), _lib.ptr(rec['ss1']), C1, _lib.ptr(dy),
                              _lib.ptr(self.g[conv.name + '.conv.weight']), _lib.ptr(ws), N, D, H, W, Cc, LRELU_SLOPE, st),
                           'wmh_conv3d_k3s2_wgrad_f16x2' if s2 else 'wmh_conv3d_k3_wgrad_f16x2')
            else:
                nws = lib.wmh_conv3d_k3_wgrad_workspace_floats(C0 + C1, Cc, N, D, H, W, conv.stride)
                ws = self._buf(wskey, (nws,))
                _lib.check(lib.wmh_conv3d_k3_wgrad(_lib.ptr(x0), _lib.ptr(rec['ss0']), C0, _lib.ptr(x1), _lib.ptr(rec['ss1']), C1, _lib.ptr(dy),
                                                 _lib.ptr(self.g[conv.name + '.conv.weight']), _lib.ptr(ws), N, D, H, W, Cc, conv.stride,
                                                 LRELU_SLOPE, st), 'wmh_conv3d_k3_wgrad')

        if not self.overlap_wgrad:
            wgrad(st, 'b.wgrad.ws')
            self._ready([wname] + rest)
            return self._block_dgrad(rec, dy, need_dx)
        # The weight gradient is off the chain IN backward -> data gradient -> IN backward of the layer below: it goes to a second
        # stream behind this layer's data gradient and runs beside the (bandwidth-bound) InstanceNorm backward of the next layer.
        self._ready(rest)
        outs = self._block_dgrad(rec, dy, need_dx)
        self._wgrad_flush()
        self._wgrad_behind(wgrad, wname)
        return outs

    def _block_dgrad(self, rec, dy, need_dx):
        lib, st = self.lib, self._st()
        conv, y = rec['conv'], rec['y']
        N, Cc = y.shape[:2]
        x0, x1 = rec['x0'], rec['x1']
        _, C0, D, H, W = x0.shape
        C1 = 0 if x1 is None else x1.shape[1]
        if not need_dx:
            return None
        if conv.stride == 2 and conv.iso and self.x3 and self.x3_s2_dgrad:   # gather form on fp16 pairs (dy is loss-scaled into fp16's range)
            assert C1 == 0
            key = ('dgrad.s2.x3', conv.name)
            hit = self._packed.get(key)
            if hit is None or hit[0] != self._version:
                wp = hit[1] if hit is not None else torch.empty(lib.wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(C0, Cc), dtype=torch.int32,
                                                                device=self.device)
                _lib.check(lib.wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device(_lib.ptr(self.p[conv.name + '.conv.weight']), C0, Cc, _lib.ptr(wp), st),
                           'wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device')
                self._packed[key] = (self._version, wp)
            wp = self._packed[key][1]
            dx = self._buf('b.dx0.%s.y' % rec['key'], (N, C0, D, H, W))
            _lib.check(lib.wmh_conv3d_k3s2_dgrad_f16x2(_lib.ptr(dy), _lib.ptr(wp), _lib.ptr(dx), N, C0, Cc, D, H, W, st), 'wmh_conv3d_k3s2_dgrad_f16x2')
            return [dx]
        if conv.stride == 2 and conv.iso:                  # gather form: 27 MFMAs per channel pair for the 8 output parities
            assert C1 == 0
            key = ('dgrad.s2', conv.name)
            hit = self._packed.get(key)
            if hit is None or hit[0] != self._version:
                wp = hit[1] if hit is not None else torch.empty(lib.wmh_conv3d_k3s2_dgrad_packed_floats(C0, Cc), dtype=torch.float32,
                                                                device=self.device)
                _lib.check(lib.wmh_conv3d_k3s2_dgrad_pack_weights_device(_lib.ptr(self.p[conv.name + '.conv.weight']), C0, Cc, _lib.ptr(wp), st),
                           'wmh_conv3d_k3s2_dgrad_pack_weights_device')
                self._packed[key] = (self._version, wp)
            wp = self._packed[key][1]
            dx = self._buf('b.dx0.%s.y' % rec['key'], (N, C0, D, H, W))
            _lib.check(lib.wmh_conv3d_k3s2_dgrad(_lib.ptr(dy), _lib.ptr(wp), _lib.ptr(dx), N, C0, Cc, D, H, W, st), 'wmh_conv3d_k3s2_dgrad')
            return [dx]
        src = dy
        if conv.stride == 2:                                 # stride (1,2,2): zeros inserted in the plane, then the stride-1 form below
            assert C1 == 0 and conv.sz == 1
            src = self._buf('b.up.%s' % rec['key'], (N, Cc, D, H, W))
            _lib.check(lib.wmh_upsample_zero_plane(_lib.ptr(dy), _lib.ptr(src), N, Cc, y.shape[3], y.shape[4], D, H, W, st),
                       'wmh_upsample_zero_plane')
        outs = []
        for part, (off, cnt) in enumerate([(0, C0)] + ([(C0, C1)] if C1 else [])):
            cfg = _lib.check(lib.wmh_conv3d_config(Cc, cnt, D, H, W, conv.kz, 1, 1), 'config')
            cfg_x3 = lib.wmh_conv3d_k3_f16x2_config(cfg, cnt, W) if (self.x3 and conv.kz == 3) else -1
            if cfg_x3 >= 0 and self._use_x3(cfg_x3, 1, Cc, 0, W):
                wf = self._pack_x3(conv, cfg_x3, 1, off, cnt)
                dx, _, _ = self._run_conv_x3('b.dx%d.%s' % (part, rec['key']), cfg_x3, src, None, None, None, wf, None, cnt, False)
            else:
                wf = self._pack(conv, cfg, 1, off, cnt)
                dx, _, _ = self._run_conv('b.dx%d.%s' % (part, rec['key']), cfg, src, None, None, None, wf, None, cnt, 1, False)
            outs.append(dx)
        return outs

    def _wgrad_behind(self, launch, wname):
        """Queue `launch(stream, workspace key)` on the weight-gradient stream behind everything the launch stream holds so far.  The
        gradient is reported ready one layer later (`_wgrad_flush`, behind the next block's data gradient), after the launch stream has
        been made to wait for it: the wait delays nothing on the chain."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self._side.wait_event(ev)
        launch(C.c_void_p(self._side.cuda_stream), 'b.wgrad.ws.side')
        done = torch.cuda.Event()
        done.record(self._side)
        self._wgrad_pending.append((done, wname))

    def _wgrad_flush(self):
        for done, wname in self._wgrad_pending:
            torch.cuda.current_stream().wait_event(done)
            self._ready([wname])
        self._wgrad_pending = []

    def _ready(self, names):
        """The gradients of these parameters are final on the launch stream: the data-parallel exchange may start."""
        if self._avg is not None:
            self._avg.ready(names)

    def backward(self, targets, group=None, force_collective=False):
        """targets: list of uint8 device tensors [N, spatial] per scale (full resolution first).  Fills the
        gradient arena and returns the device scalar holding the loss.  Under torch.distributed the arena is averaged
        over the ranks on the way: each parameter group's all-reduce starts as soon as its gradient is complete and
        runs next to the rest of the backward pass (distributed.OverlappedGradientAverager; with one process it only
        checks that every parameter's gradient was produced exactly once)."""
        from .distributed import OverlappedGradientAverager
        if self._avg is None or self._avg.group is not group or getattr(self._avg, 'forced', False) != bool(force_collective):
            self._avg = OverlappedGradientAverager(self.grads, self.grad_ranges, group, min_elems=2 * 1024 * 1024,   # >= 8 MB per message
                                                   force_collective=force_collective)
            self._avg.forced = bool(force_collective)
        self._avg.begin()
        self._wgrad_pending = []                             # (a backward pass that died half way must not report its gradients into this one)
        self._side_retired = []                              # (the previous pass ended with the two streams joined: its retired workspaces may go)
        self._update_loss_scale()
        lib, st, tape = self.lib, self._st(), self.tape
        nd = len(self.dec)
        self.loss_acc.zero_()
        dls = [None] * nd
        for s in range(nd):                                  # scale s <-> decoder stage u = nd-1-s
            u = nd - 1 - s
            w = float(self.ds_weights[s])
            lg = tape['dec'][u]['logits']
            N = lg.shape[0]
            S = int(np.prod(lg.shape[2:]))
            if w == 0.0:
                self.g[self.dec[u]['seg'] + '.weight'].zero_()
                self._ready([self.dec[u]['seg'] + '.weight'])
                continue
            dl = self._buf('b.dl%d' % u, lg.shape)
            _lib.check(lib.wmh_dice_ce_loss_fwd_bwd(_lib.ptr(lg), _lib.ptr(targets[s]), N, self.K, S, w * self.loss_scale, 1e-5, _lib.ptr(dl),
                                                    _lib.ptr(self.loss_acc), _lib.ptr(self.ws_d), st), 'wmh_dice_ce_loss_fwd_bwd')
            dls[u] = dl
        g_next = None                                        # gradient arriving from the stage above (via its convT)
        gskip = {}
        for u in range(nd - 1, -1, -1):
            rec = tape['dec'][u]
            seg = self.dec[u]['seg'] + '.weight'
            (gA0,) = self._block_bwd(rec['r1'], g_next, None, dls[u], seg if dls[u] is not None else None, True)
            d_up, d_skip = self._block_bwd(rec['r0'], gA0, None, None, None, True)
            gskip[u] = d_skip
            x_in, ss_in = rec['tu_in']
            N, Cin, D, H, W = x_in.shape
            tu = self.dec[u]['tu'] + '.weight'
            cskip = self.dec[u]['cskip']
            # dW on the matrix cores; dX as a stride-2 convolution of dY with the 2-window embedded in the 3-window
            kt = self.dec[u]['kt']
            nws = lib.wmh_convtranspose3d_wgrad_workspace_floats(Cin, cskip, N, D, H, W, kt)
            if not self.overlap_wgrad:
                ws = self._buf('b.wgrad.ws', (max(nws, 1),))
                _lib.check(lib.wmh_convtranspose3d_wgrad(_lib.ptr(x_in), _lib.ptr(ss_in), _lib.ptr(d_up), _lib.ptr(self.g[tu]), _lib.ptr(ws),
                                                         N, Cin, cskip, D, H, W, kt, LRELU_SLOPE, st), 'wmh_convtranspose3d_wgrad')
                self._ready([tu])
            # (1,2,2): the window embedded in a (1,3,3) one, stride (1,2,2)
            cfg = _lib.check(lib.wmh_conv3d_config(cskip, Cin, kt * D, 2 * H, 2 * W, 3 if kt == 2 else 1, kt, 2), 'config')
            cfg_x3 = lib.wmh_conv3d_k3_f16x2_config(cfg, Cin, 2 * W) if (self.x3 and kt == 2 and self.x3_tu_dgrad) else -1
            if cfg_x3 in (1, 3, 5, 25, 26) and bool(lib.wmh_conv3d_k3_f16x2_supported(cfg_x3)):     # the strided pair kernels on the embedded image
                key = ('tu.dgrad.x3', u, cfg_x3)
                hit = self._packed.get(key)
                self._x3_used[key] = self._version          # (as _pack_x3: an image nobody records is dropped by repack_x3_batch)
                if hit is None or hit[0] != self._version:
                    wfx = hit[1] if hit is not None else torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg_x3, cskip, Cin), dtype=torch.int16,
                                                                     device=self.device)
                    self._pack_x3_launch(key, (cfg_x3, self.p[tu], cskip, Cin, 0, cskip, 2, wfx))
                wfx = self._packed[key][1]
                g_in, _, _ = self._run_conv_x3('b.tu%d.dx' % u, cfg_x3, d_up, None, None, None, wfx, None, Cin, False, 2)
            else:
                wf = self._pack_tu_dgrad(u, self.p[tu], cfg, Cin, cskip)
                g_in, _, _ = self._run_conv('b.tu%d.dx' % u, cfg, d_up, None, None, None, wf, None, Cin, 2, False, kt)
            g_next = g_in
            if self.overlap_wgrad:                           # behind the data gradient, like the convolutions' weight gradients
                def tu_wgrad(sst, wskey, x_in=x_in, ss_in=ss_in, d_up=d_up, tu=tu, nws=nws, dims=(N, Cin, cskip, D, H, W, kt)):
                    ws = self._buf(wskey + '.tu', (max(nws, 1),))
                    _lib.check(lib.wmh_convtranspose3d_wgrad(_lib.ptr(x_in), _lib.ptr(ss_in), _lib.ptr(d_up), _lib.ptr(self.g[tu]), _lib.ptr(ws),
                                                             *dims, LRELU_SLOPE, sst), 'wmh_convtranspose3d_wgrad')
                self._wgrad_behind(tu_wgrad, tu)
        # encoder: bottleneck back to the first layer; stage outputs also receive the decoder's skip gradient
        nenc = len(self.enc)
        g = g_next
        for i in range(nenc - 1, -1, -1):
            g2 = None
            if i % 2 == 1 and i < nenc - 2:                  # output of encoder stage d = i//2 -> skip of decoder u = nd-1-d
                g2 = gskip[nd - 1 - i // 2]
            outs = self._block_bwd(tape['enc'][i], g, g2, None, None, i > 0)
            g = outs[0] if outs else None
        self._wgrad_flush()
        self._avg.finish()
        if self.loss_scale != 1.0:                           # exact: power of two
            self.grads.mul_(1.0 / self.loss_scale)
            self.loss_acc.mul_(1.0 / self.loss_scale)
        return self.loss_acc

    # ---- optimizer -----------------------------------------------------------------------------------------------
    def _update_loss_scale(self):
        """Looks at the gradient norm of the PREVIOUS optimizer step (copied to pinned memory behind it): a non-finite one means the
        device skipped that update; halve the scale for the iterations to come.  Called before a backward pass picks the scale."""
        if self._norm_event is None:
            return
        self._norm_event.synchronize()                       # long done: a whole forward pass has been queued since
        self._norm_event = None
        if bool(np.isfinite(self._norm_host[0].item())):
            self._clean_steps += 1
            if self._clean_steps >= self.growth_interval and self.loss_scale < self.max_loss_scale:
                self.loss_scale, self._clean_steps = min(self.max_loss_scale, self.loss_scale * 2.0), 0
        else:
            self.skipped_steps += 1
            self._clean_steps = 0
            if self.loss_scale != 1.0:
                self.loss_scale = max(self.min_loss_scale, self.loss_scale * 0.5)

    def optimizer_step(self, lr):
        st = self._st()
        _lib.check(self.lib.wmh_grad_clip_coef(_lib.ptr(self.grads), self.nparams, self.clip, _lib.ptr(self.clip_out), _lib.ptr(self.ws_d),
                                               st), 'wmh_grad_clip_coef')
        _lib.check(self.lib.wmh_sgd_nesterov_step(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.mom), self.nparams, lr,
                                                  self.momentum, self.weight_decay, _lib.ptr(self.clip_out), 1 if self.first_step else 0,
                                                  st), 'wmh_sgd_nesterov_step')
        self.first_step = False                              # (a skipped first step leaves the zero momentum: same arithmetic either way)
        self._version += 1                                   # packed weight images are stale
        self._norm_host.copy_(self.clip_out, non_blocking=True)
        self._norm_event = torch.cuda.Event()
        self._norm_event.record(torch.cuda.current_stream())

    def allreduce_grads(self, group=None):
        """The exchange without overlap, for gradients that were not produced by backward(): one bucketed RCCL all-reduce
        of the flat arena, then the mean.  backward() itself already averages (overlapped); do not call both."""
        from .distributed import average_gradients
        average_gradients(self.grads, group, bucket_elems=8 * 1024 * 1024)    # 32 MB buckets over xGMI

    def run_iteration(self, data, targets, lr, do_backprop=True, group=None):
        """nnUNetTrainerV2.run_iteration.  data [N,C,...] device fp32, targets = per-scale uint8 label tensors."""
        self.forward(data)
        loss = self.backward(targets, group) if do_backprop else self.loss_only(targets)   # data parallel: averaged on the way
        if do_backprop:
            self.optimizer_step(lr)
        return loss

    def online_eval(self, target_full):
        """nnUNetTrainerV2.run_online_evaluation on the batch the last forward() saw: the hard tp / fp / fn counts of every foreground class at
        full resolution, ADDED to `self.eval_counts` (int64 [3 (K - 1)], zero it where an epoch's validation starts).  What `model_best` is
        decided by (criterion.py)."""
        if getattr(self, 'eval_counts', None) is None:
            self.eval_counts = torch.zeros(3 * (self.K - 1), dtype=torch.int64, device=self.device)
        lg = self.tape['dec'][len(self.dec) - 1]['logits']
        _lib.check(self.lib.wmh_online_eval_counts(_lib.ptr(lg), _lib.ptr(target_full), lg.shape[0], self.K, int(np.prod(lg.shape[2:])),
                                                   _lib.ptr(self.eval_counts), self._st()), 'wmh_online_eval_counts')
        return self.eval_counts

    def loss_only(self, targets):
        nd = len(self.dec)
        self.loss_acc.zero_()
        for s in range(nd):
            w = float(self.ds_weights[s])
            if w == 0.0:
                continue
            lg = self.tape['dec'][nd - 1 - s]['logits']
            _lib.check(self.lib.wmh_dice_ce_loss_fwd_bwd(_lib.ptr(lg), _lib.ptr(targets[s]), lg.shape[0], self.K,
                                                         int(np.prod(lg.shape[2:])), w, 1e-5, None, _lib.ptr(self.loss_acc),
                                                         _lib.ptr(self.ws_d), self._st()), 'loss')
        return self.loss_acc


def ds_pick_indices(in_size, factor):
    """Voxels nnU-Net's order-0 `resize_segmentation` [batchgenerators, via DownsampleSegForDSTransform2; not in tree] keeps
    along one axis when it shrinks it by `factor`: out = round(in / factor) (half to even), output voxel o reads input
    coordinate (o + 0.5) * in / out - 0.5 rounded to nearest (skimage resize / scipy zoom with grid_mode) = floor((o + 0.5) *
    in / out), clipped.  Factor 2 keeps voxels 1, 3, 5, ...; factor 4 keeps 2, 6, 10, ..."""
    out = int(np.round(in_size / float(factor)))
    o = np.arange(out, dtype=np.float64)
    return np.clip(np.floor((o + 0.5) * in_size / out).astype(np.int64), 0, in_size - 1)


_DS_INDEX_CACHE = {}


def _ds_index_tensor(in_size, factor, device):
    """The pick indices as a device tensor, built once per (length, factor, device): rebuilding them per iteration was a dozen
    pageable, host-synchronous H2D copies per training step."""
    key = (int(in_size), int(factor), str(device))
    t = _DS_INDEX_CACHE.get(key)
    if t is None:
        t = _DS_INDEX_CACHE[key] = torch.from_numpy(ds_pick_indices(in_size, factor)).to(device)
    return t


def downsample_targets(seg, num_scales, pool_op_kernel_sizes=None):
    """Deep-supervision targets (nnUNetTrainerV2: scales 1 / cumprod(pooling kernels), order 0).  seg: uint8 tensor
    [N, X, Y, Z] on any device; returns one [N, voxels] tensor per scale."""
    pools = pool_op_kernel_sizes or [[2, 2, 2]] * max(0, num_scales - 1)
    out, cum = [seg.contiguous().view(seg.shape[0], -1)], [1, 1, 1]
    for s in range(1, num_scales):
        cum = [c * int(p) for c, p in zip(cum, pools[s - 1])]
        t = seg
        for ax in range(3):
            if cum[ax] != 1:
                t = t.index_select(1 + ax, _ds_index_tensor(seg.shape[1 + ax], cum[ax], seg.device))
        out.append(t.contiguous().view(seg.shape[0], -1))
    return out
