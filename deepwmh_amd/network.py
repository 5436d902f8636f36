"""Generic_UNet forward on the HIP kernels (host orchestration; torch is the HBM allocator only).

Mirrors nnU-Net `Generic_UNet.forward` [nnU-Net, not in tree; generic_UNet.py] layer for layer:
encoder Conv-IN-LReLU x2 per stage (first conv of a stage strided), bottleneck, decoder
ConvTranspose3d -> (concat-free) Conv-IN-LReLU x2, segmentation heads.  InstanceNorm+LeakyReLU of a
layer is never materialised: the conv epilogue emits statistics, `wmh_instnorm_finalize` turns
them into per-(n,c) {scale, shift}, and every consumer applies them while loading.
Checkpoint keys are the `state_dict` keys of the reference's `model_best.model`
(/root/reference/deepwmh/main/install_model.py:85-106 pins where that file lives).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .plan import UNetPlan

LRELU_SLOPE = 1e-2
IN_EPS = 1e-5


def _np(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class _ConvLayer:
    """One Conv3d(3x3x3)+InstanceNorm layer: packed weights per kernel configuration."""

    def __init__(self, lib, sd, prefix, cin, cout, stride, device, kz=3, sz=None):
        self.lib, self.cin, self.cout, self.stride, self.device = lib, cin, cout, stride, device
        self.kz, self.sz = kz, (stride if sz is None else sz)
        self.w_host = _np(sd[prefix + '.conv.weight'])
        assert self.w_host.shape == (cout, cin, kz, 3, 3), (prefix, self.w_host.shape)
        self.bias = torch.from_numpy(_np(sd[prefix + '.conv.bias'])).to(device)
        self.gamma = torch.from_numpy(_np(sd[prefix + '.instnorm.weight'])).to(device)
        self.beta = torch.from_numpy(_np(sd[prefix + '.instnorm.bias'])).to(device)
        self._packed = {}
        self._packed_bf16 = {}

    def packed_bf16(self, cfg):
        if cfg not in self._packed_bf16:
            n = self.lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, self.cin, self.cout)
            wf = np.empty(n, dtype=np.uint16)
            _lib.check(self.lib.wmh_conv3d_k3_pack_weights_bf16(
                cfg, self.w_host.ctypes.data_as(C.c_void_p), self.cin, self.cout,
                wf.ctypes.data_as(C.c_void_p)), 'wmh_conv3d_k3_pack_weights_bf16')
            self._packed_bf16[cfg] = torch.from_numpy(wf.view(np.int16)).to(self.device)
        return self._packed_bf16[cfg]

    def packed_f16x2(self, cfg):
        """The layer's f16x2 weight image.  Built on the DEVICE since round 6 (the packer the trainer uses every step; bit-identical to the host packer:
        tests/test_gpu_f16x2.py): the host loops took ~0.4 s for the BASELINE network -- most of the first forward of a fresh `nnUNet_predict`
        process, which DeepWMH_predict starts once per case.  The device packer does not look at the values, so the range check the host packer
        makes (finite, |256 w| < 65000) is made here; weights outside it go to the host packer, which refuses them with the reason."""
        key = ('x3', cfg)
        if key not in self._packed_bf16:
            n = self.lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, self.cin, self.cout)
            amax = float(np.abs(self.w_host).max()) if self.w_host.size else 0.0
            if self.kz == 3 and np.isfinite(amax) and amax * 256.0 < 65000.0:
                wd = torch.from_numpy(np.ascontiguousarray(self.w_host.reshape(self.cout, self.cin, -1))).to(self.device)
                wf = torch.empty(n, dtype=torch.int16, device=self.device)
                _lib.check(self.lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, _lib.ptr(wd), self.cin, self.cout, 0, self.cin, 0, _lib.ptr(wf),
                                                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wmh_conv3d_k3_pack_weights_f16x2_device')
                wd.record_stream(torch.cuda.current_stream())          # freed when the launch has read it
                self._packed_bf16[key] = wf
            else:
                wf = np.empty(n, dtype=np.uint16)
                _lib.check(self.lib.wmh_conv3d_k3_pack_weights_f16x2(
                    cfg, self.w_host.ctypes.data_as(C.c_void_p), self.cin, self.cout,
                    wf.ctypes.data_as(C.c_void_p)), 'wmh_conv3d_k3_pack_weights_f16x2')
                self._packed_bf16[key] = torch.from_numpy(wf.view(np.int16)).to(self.device)
        return self._packed_bf16[key]

    def packed_f16x2_slice(self, cfg, ci_off, ci_cnt):
        """The f16x2 image of the input-channel sub-range [ci_off, ci_off + ci_cnt) (a decoder stage's skip half), built on the device."""
        key = ('x3', cfg, ci_off, ci_cnt)
        if key not in self._packed_bf16:
            wd = torch.from_numpy(self.w_host.reshape(self.cout, self.cin, -1)).to(self.device)
            wf = torch.empty(self.lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, ci_cnt, self.cout), dtype=torch.int16, device=self.device)
            _lib.check(self.lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, _lib.ptr(wd), self.cin, self.cout, ci_off, ci_cnt, 0, _lib.ptr(wf),
                                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wmh_conv3d_k3_pack_weights_f16x2_device')
            torch.cuda.current_stream().synchronize()          # wd goes out of scope
            self._packed_bf16[key] = wf
        return self._packed_bf16[key]

    def packed(self, cfg):
        if cfg not in self._packed:
            n = self.lib.wmh_conv3d_k3_packed_floats(cfg, self.cin, self.cout)
            wf = np.empty(n, dtype=np.float32)
            _lib.check(self.lib.wmh_conv3d_k3_pack_weights(
                cfg, self.w_host.ctypes.data_as(C.c_void_p), self.cin, self.cout,
                wf.ctypes.data_as(C.c_void_p)), 'wmh_conv3d_k3_pack_weights')
            self._packed[cfg] = torch.from_numpy(wf).to(self.device)
        return self._packed[cfg]


class UNetEngine:
    """Holds the weights in HBM and runs the forward graph on a HIP stream."""

    def __init__(self, plan: UNetPlan, state_dict, device='cuda:0', precision='fp32', storage='fp32', fuse_up=None):
        """precision 'fp32' (exact-fp32 matrix instruction everywhere) or 'bf16' (BASELINE.json config 3: the
        stride-1 convolutions of the >= 16-wide levels multiply bf16-rounded operands with fp32 accumulation;
        InstanceNorm statistics, heads and softmax stay fp32).  storage 'bf16' (with precision 'bf16' only) also
        keeps the five full-resolution activation tensors -- where the HBM bytes are -- as bf16, like the fp16
        tensors of nnU-Net's autocast predict; everything below the first pooling stays fp32."""
        if precision not in ('fp32', 'bf16', 'f16x2'):
            raise ValueError('precision must be fp32, bf16 or f16x2')
        if storage not in ('fp32', 'bf16', 'bf16_ncdhw') or (storage != 'fp32' and precision != 'bf16'):
            raise ValueError("storage must be 'fp32', or 'bf16' / 'bf16_ncdhw' together with precision='bf16'")
        self.precision = precision
        self.storage = storage
        # f16x2: a decoder stage's transposed convolution is composed with the `up` half of the convolution behind it where the geometry allows
        # (csrc/conv3d_up_f16x2.hip): no `up` tensor.  fuse_up=False (WMH_FUSE_UP=0) keeps the two-kernel form (A/B runs, tests).
        import os
        self.fuse_up = (os.environ.get('WMH_FUSE_UP', '1') != '0') if fuse_up is None else bool(fuse_up)
        plan.check_supported()
        self.plan = plan
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('UNetEngine needs a HIP device (got %s); there is no CPU path' % device)
        self.lib = _lib.load()
        _lib.check(self.lib.wmh_device_check(self.device.index or 0), 'wmh_device_check')
        sd = state_dict
        dev = self.device
        self.enc = [_ConvLayer(self.lib, sd, s.name, s.cin, s.cout, s.stride, dev, s.kz, s.sz) for s in plan.encoder_convs()]
        self.dec = []
        for st in plan.decoder_stages():
            name, cdown, cskip = st['tu']
            kt = st['tu_kt']
            tu_host = _np(sd[name + '.weight'])
            assert tuple(tu_host.shape) == (cdown, cskip, kt, 2, 2)
            tu_w = torch.from_numpy(tu_host).to(dev)
            wf = np.empty(self.lib.wmh_convtranspose3d_packed_floats(cdown, cskip, kt), dtype=np.float32)
            _lib.check(self.lib.wmh_convtranspose3d_pack_weights(tu_host.ctypes.data_as(C.c_void_p), cdown, cskip, kt,
                                                                 wf.ctypes.data_as(C.c_void_p)), 'convT pack')
            tu_wf = torch.from_numpy(wf).to(dev)
            tu_wf_x3 = None
            if precision == 'f16x2':                       # pair image of the same weights (split on the device)
                tu_wf_x3 = torch.empty(self.lib.wmh_convtranspose3d_packed_dwords_f16x2(cdown, cskip, kt), dtype=torch.int32, device=dev)
                _lib.check(self.lib.wmh_convtranspose3d_pack_weights_f16x2_device(_lib.ptr(tu_w), cdown, cskip, kt, _lib.ptr(tu_wf_x3), None),
                           'wmh_convtranspose3d_pack_weights_f16x2_device')
            convs = [_ConvLayer(self.lib, sd, c.name, c.cin, c.cout, 1, dev, c.kz, c.sz) for c in st['convs']]
            seg_w = torch.from_numpy(_np(sd[st['seg'] + '.weight']).reshape(plan.num_classes, cskip)).contiguous().to(dev)
            self.dec.append(dict(tu_w=tu_w, tu_wf=tu_wf, tu_wf_x3=tu_wf_x3, tu_host=tu_host, kt=kt, cdown=cdown, cskip=cskip, convs=convs, seg_w=seg_w))
        if storage == 'bf16_ncdhw':
            self._check_storage_supported()
        self.ncl = 0                                   # resolution levels held as channels-last bf16 tensors
        if storage == 'bf16':
            self._setup_channels_last(sd)
        self._bufs = {}
        # optional live kernel timing (bench.py): conv configs to bracket with HIP events on the launch stream
        self.timed_cfgs = set()
        self.timings = []          # (cfg, start_event, end_event, algorithmic_flop, algorithmic_bytes)
        # optional accounting of the algorithmic work queued (bench.py): {'flop': .., 'bytes': ..} or None.  Bytes follow
        # every layer reads its input once and writes its output once, plus its weights.
        self.acct = None

    def clone(self):
        """A second engine on the same weights (packed images included) with activation buffers of its own: `nnUNet_predict` keeps two cases
        in flight on two streams when a case is too short to hide the host work around it (--disable_tta)."""
        import copy
        c = copy.copy(self)
        c._bufs, c.timed_cfgs, c.timings, c.acct = {}, set(), [], None
        return c

    def _check_storage_supported(self):
        """storage='bf16' is served by specific kernel configurations (include/deepwmh_amd.h, the *_st calls): say so
        when the engine is built, not in the middle of a prediction."""
        plan = self.plan
        D, H, W = (int(v) for v in plan.patch_size)
        e0, e1, e2 = self.enc[0], self.enc[1], self.enc[2]
        cfg = lambda l, d, h, w: self.lib.wmh_conv3d_config(l.cin, l.cout, d, h, w, l.kz, l.sz, l.stride)
        problems = []
        if W % 4 != 0:
            problems.append('the last patch axis must be a multiple of 4 (is %d)' % W)
        if cfg(e0, D, H, W) != 8:
            problems.append('the first layer must have 1-2 input channels, a 3x3x3 kernel and rows of >= 24 voxels')
        if cfg(e1, D, H, W) != 0 or not self.lib.wmh_conv3d_k3_bf16_supported(0):
            problems.append('the full-resolution level must have rows of >= 24 voxels and 3x3x3 kernels')
        if cfg(e2, D, H, W) != 1:
            problems.append('the first pooling must be 2x2x2 onto rows of >= 24 voxels (patch rows of >= 47)')
        if e1.cout % 16 != 0:
            problems.append('the full-resolution level must have a multiple of 16 feature maps (has %d)' % e1.cout)
        last = self.dec[-1]
        if last['kt'] != 2 or any(c.kz != 3 for c in last['convs']):
            problems.append('the last decoder stage must be isotropic (2x2x2 up-sampling, 3x3x3 kernels)')
        if problems:
            raise ValueError("storage='bf16' is not available for this plan: " + '; '.join(problems))

    # ---- buffers -------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype=torch.float32):
        t = self._bufs.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t[:n].view(*shape)

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- layers --------------------------------------------------------------------------------
    def _conv(self, key, layer, x0, ss0, x1=None, ss1=None, out16=False):
        """out16: write y as a bf16 tensor (storage='bf16', full-resolution level).  bf16 sources are recognised by dtype.
        Combinations the kernels do not serve fail in the library (WMH_EINVAL): there is no conversion fallback."""
        N, C0, D, H, W = x0.shape
        in16 = x0.dtype == torch.bfloat16
        if x1 is not None and (x1.dtype == torch.bfloat16) != in16:
            raise ValueError('both sources of a convolution must use the same storage format')
        C1 = 0 if x1 is None else x1.shape[1]
        assert C0 + C1 == layer.cin
        s = layer.stride
        Do, Ho, Wo = (D - 1) // layer.sz + 1, (H - 1) // s + 1, (W - 1) // s + 1
        cfg = _lib.check(self.lib.wmh_conv3d_config(layer.cin, layer.cout, D, H, W, layer.kz, layer.sz, s), 'wmh_conv3d_config')
        # bf16 operands: every 3x3x3 configuration the library has a bf16 kernel for (stride 1: rows of >= 8 voxels; stride 2:
        # output rows of 8 or 16 voxels; the stride-2 convolutions with longer rows are channels-last or stay fp32)
        plain = not (in16 or out16)
        use_bf16 = (self.precision == 'bf16' and layer.kz == 3 and layer.sz == s and (C1 == 0 or C0 % 16 == 0)
                    and bool(self.lib.wmh_conv3d_k3_bf16_supported(cfg)) and (plain or (cfg == 0 and s == 1)))
        # 'f16x2': the 3x3x3 convolutions that have a 16-bit-operand kernel (stride 1: rows of >= 8 voxels; stride 2: output rows of 8 or 16)
        # form their products from fp16 pairs (x = hi + lo, 21-22 significant bits per operand; the stride-1 kernels form three of the four partial products, lo x lo -- 2^-22 of a product -- being left out; fp32
        # accumulation): results at the exact-fp32 kernel's error level at 2-2.7x its speed.  Everything else is the exact-fp32 path.
        cfg_x = self.lib.wmh_conv3d_k3_f16x2_config(cfg, layer.cout, W)
        use_x3 = (self.precision == 'f16x2' and layer.kz == 3 and layer.sz == s and (C1 == 0 or C0 % 8 == 0)
                  and bool(self.lib.wmh_conv3d_k3_f16x2_supported(cfg_x)) and (cfg_x in (3, 4, 5) or W % 4 == 0))      # (1: the strided wave-specialised kernel loads aligned quads too)
        if use_x3:
            cfg = cfg_x
        if use_bf16 and not (in16 or out16):       # wide tile for >= 64 output channels (bf16 kernel only)
            cfg = self.lib.wmh_conv3d_k3_bf16_config(cfg, layer.cout, W)
        if use_x3:
            npart = _lib.check(self.lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, Do, Ho, Wo), 'wmh_conv3d_k3_num_partials_f16x2')
        elif use_bf16:
            npart = _lib.check(self.lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, Do, Ho, Wo), 'wmh_conv3d_k3_num_partials_bf16')
        else:
            npart = _lib.check(self.lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo), 'wmh_conv3d_k3_num_partials')
        y = self._buf(key + ('.y16' if out16 else '.y'), (N, layer.cout, Do, Ho, Wo), torch.bfloat16 if out16 else torch.float32)
        part = self._buf(key + '.part', (N, layer.cout, npart, 4))
        ss = self._buf(key + '.ss', (N, layer.cout, 2))
        if use_x3:
            nws = self.lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, layer.cin, N, D, H, W, layer.cout)
        elif use_bf16:
            nws = self.lib.wmh_conv3d_k3_bf16_workspace_floats(cfg, layer.cin, N, D, H, W, layer.cout)
        else:
            nws = self.lib.wmh_conv3d_k3_workspace_floats(cfg, layer.cin, N, D, H, W, layer.cout, s)
        ws = self._buf('splitk.ws', (nws,)) if nws else None      # shared: layers run one after another
        st = self._stream()
        wf = layer.packed_f16x2(cfg) if use_x3 else layer.packed_bf16(cfg) if use_bf16 else layer.packed(cfg)
        timed = cfg in self.timed_cfgs
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if use_x3:
            _lib.check(self.lib.wmh_conv3d_k3_fwd_f16x2(
                cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1,
                _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws),
                N, D, H, W, layer.cout, LRELU_SLOPE, st), 'wmh_conv3d_k3_fwd_f16x2')
        elif (in16 or out16) and use_bf16:
            _lib.check(self.lib.wmh_conv3d_k3_fwd_bf16_st(
                cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1,
                _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part),
                N, D, H, W, layer.cout, LRELU_SLOPE, int(in16), int(out16), st), 'wmh_conv3d_k3_fwd_bf16_st')
        elif in16 or out16:
            _lib.check(self.lib.wmh_conv3d_k3_fwd_st(
                cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1,
                _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part),
                N, D, H, W, layer.cout, s, LRELU_SLOPE, int(in16), int(out16), st), 'wmh_conv3d_k3_fwd_st')
        elif use_bf16:
            _lib.check(self.lib.wmh_conv3d_k3_fwd_bf16_splitk(
                cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1,
                _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws),
                N, D, H, W, layer.cout, LRELU_SLOPE, st), 'wmh_conv3d_k3_fwd_bf16_splitk')
        else:
            _lib.check(self.lib.wmh_conv3d_k3_fwd(
                cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1,
                _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws),
                N, D, H, W, layer.cout, s, LRELU_SLOPE, st), 'wmh_conv3d_k3_fwd')
        if timed or self.acct is not None:
            vox = N * Do * Ho * Wo
            flop = 2.0 * layer.cin * layer.cout * 9 * layer.kz * vox
            nbytes = ((2.0 if in16 else 4.0) * N * layer.cin * D * H * W + (2.0 if out16 else 4.0) * layer.cout * vox
                      + (2.0 if use_bf16 else 4.0) * layer.cout * layer.cin * 9 * layer.kz)
            if timed:
                e1.record()
                self.timings.append((cfg, e0, e1, flop, nbytes))
            if self.acct is not None:
                self.acct['flop'] += flop
                self.acct['bytes'] += nbytes
        _lib.check(self.lib.wmh_instnorm_finalize(
            _lib.ptr(part), npart, _lib.ptr(layer.gamma), _lib.ptr(layer.beta), IN_EPS, _lib.ptr(ss), None,
            N, layer.cout, st), 'wmh_instnorm_finalize')
        return y, ss

    def _up_image(self, st):
        """Composed weight image of a decoder stage (transposed convolution x the `up` input channels of its first convolution), built once."""
        if 'up_wimg' not in st:
            layer = st['convs'][0]
            Cu, Cl = st['cskip'], st['cdown']              # tu: cdown -> cskip channels; cat((up, skip)): the first cskip inputs of the convolution
            wimg = np.zeros(self.lib.wmh_conv3d_up_packed_bytes_f16x2(Cl, layer.cout), np.uint8)
            w3 = np.ascontiguousarray(layer.w_host.reshape(layer.cout, layer.cin, 27))
            wt = np.ascontiguousarray(st['tu_host'].reshape(Cl, Cu, 8))
            rc = self.lib.wmh_conv3d_up_pack_weights_f16x2(w3.ctypes.data_as(C.c_void_p), layer.cin, Cu, wt.ctypes.data_as(C.c_void_p), Cl,
                                                           layer.cout, wimg.ctypes.data_as(C.c_void_p))
            # a composed weight beyond fp16's range (or not finite): this stage is "not fusable" for good -- transposed convolution + two-source
            # convolution, whose own packers decide about the two layers' weights
            st['up_wimg'] = torch.from_numpy(wimg).to(self.device) if rc == 0 else None
        return st['up_wimg']

    def _dec_first_fused(self, key, st, low, low_ss, skip, skip_ss):
        """First convolution of a decoder stage WITHOUT the `up` tensor: the pair kernel on the skip channels (y = bias + conv(skip)), then the
        composed transposed-convolution x convolution kernel adds the rest and takes the moments.  Returns (y, ss) or None where the
        geometry is not served (then the caller runs the transposed convolution and the two-source convolution)."""
        layer = st['convs'][0]
        lib = self.lib
        N, Cl, Dl, Hl, Wl = low.shape
        D, H, W = 2 * Dl, 2 * Hl, 2 * Wl
        Cs = skip.shape[1]
        Cu = layer.cin - Cs
        if (self.precision != 'f16x2' or not self.fuse_up or st['kt'] != 2 or layer.kz != 3 or layer.sz != 1 or Cu != st['cskip'] or W % 4 != 0
                or tuple(skip.shape[2:]) != (D, H, W) or not lib.wmh_conv3d_up_supported_f16x2(Cl, Cu, layer.cout, D, H, W)):
            return None
        cfg = lib.wmh_conv3d_k3_bf16_config(_lib.check(lib.wmh_conv3d_config(Cs, layer.cout, D, H, W, 3, 1, 1), 'wmh_conv3d_config'), layer.cout, W)
        if cfg not in (0, 2, 21) or not lib.wmh_conv3d_k3_f16x2_supported(cfg):
            return None
        wimg = self._up_image(st)
        if wimg is None:
            return None
        wf = layer.packed_f16x2_slice(cfg, Cu, Cs)
        npart = _lib.check(lib.wmh_conv3d_up_num_partials_f16x2(D, H, W), 'wmh_conv3d_up_num_partials_f16x2')
        y = self._buf(key + '.y', (N, layer.cout, D, H, W))
        part = self._buf(key + '.part', (N, layer.cout, npart, 4))
        ss = self._buf(key + '.ss', (N, layer.cout, 2))
        st_ = self._stream()
        timed = cfg in self.timed_cfgs
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, _lib.ptr(skip), _lib.ptr(skip_ss), Cs, None, None, 0, _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y),
                                               None, None, N, D, H, W, layer.cout, LRELU_SLOPE, st_), 'wmh_conv3d_k3_fwd_f16x2 (skip half)')
        vox = N * D * H * W
        if timed:
            e1.record()
            self.timings.append((cfg, e0, e1, 2.0 * Cs * layer.cout * 27 * vox, 4.0 * N * Cs * D * H * W + 4.0 * layer.cout * vox + 4.0 * layer.cout * Cs * 27))
        _lib.check(lib.wmh_conv3d_up_fwd_f16x2(_lib.ptr(low), _lib.ptr(low_ss), Cl, _lib.ptr(wimg), _lib.ptr(y), _lib.ptr(part), N, D, H, W, layer.cout,
                                               LRELU_SLOPE, 1, st_), 'wmh_conv3d_up_fwd_f16x2')
        if self.acct is not None:
            # the reference's work for this stage: transposed convolution + 27-tap convolution over all input channels;
            # bytes: both sources once, the output once, the weights (no `up` tensor: it is not part of what this path moves)
            self.acct['flop'] += 2.0 * Cl * Cu * N * Dl * Hl * Wl * 8 + 2.0 * layer.cin * layer.cout * 27 * vox
            self.acct['bytes'] += 4.0 * low.numel() + 4.0 * skip.numel() + 4.0 * layer.cout * vox + 4.0 * (layer.cout * layer.cin * 27 + Cl * Cu * 8)
        _lib.check(lib.wmh_instnorm_finalize(_lib.ptr(part), npart, _lib.ptr(layer.gamma), _lib.ptr(layer.beta), IN_EPS, _lib.ptr(ss), None,
                                             N, layer.cout, st_), 'wmh_instnorm_finalize')
        return y, ss

    def _convT(self, key, w, wf, x, ss, cout, kt=2, out16=False, wf_x3=None):
        N, Cin, D, H, W = x.shape
        if self.acct is not None:
            self.acct['flop'] += 2.0 * Cin * cout * N * D * H * W * 4 * kt
            self.acct['bytes'] += ((2.0 if x.dtype == torch.bfloat16 else 4.0) * x.numel()
                                   + (2.0 if out16 else 4.0) * N * cout * D * H * W * 4 * kt + 4.0 * Cin * cout * 4 * kt)
        if out16:
            if (D * H * W) % 4 != 0:
                raise ValueError("storage='bf16' needs D*H*W % 4 == 0 below the full-resolution level")
            y = self._buf(key + '.y16', (N, cout, kt * D, 2 * H, 2 * W), torch.bfloat16)
            _lib.check(self.lib.wmh_convtranspose3d_mfma_fwd_st(
                _lib.ptr(x), _lib.ptr(ss), _lib.ptr(wf), _lib.ptr(y), N, Cin, cout, D, H, W, kt, LRELU_SLOPE, 1,
                self._stream()), 'wmh_convtranspose3d_mfma_fwd_st')
            return y
        y = self._buf(key + '.y', (N, cout, kt * D, 2 * H, 2 * W))
        if wf_x3 is not None and x.dtype == torch.float32:      # 'f16x2': products from fp16 pairs, bound by its tensors instead of the fp32 MFMA
            _lib.check(self.lib.wmh_convtranspose3d_fwd_f16x2(
                _lib.ptr(x), _lib.ptr(ss), _lib.ptr(wf_x3), _lib.ptr(y), N, Cin, cout, D, H, W, kt, LRELU_SLOPE,
                self._stream()), 'wmh_convtranspose3d_fwd_f16x2')
            return y
        if (D * H * W) % 4 == 0:           # matrix-core kernel (16-byte staging loads)
            _lib.check(self.lib.wmh_convtranspose3d_mfma_fwd(
                _lib.ptr(x), _lib.ptr(ss), _lib.ptr(wf), _lib.ptr(y), N, Cin, cout, D, H, W, kt, LRELU_SLOPE,
                self._stream()), 'wmh_convtranspose3d_mfma_fwd')
        elif kt != 2:
            raise NotImplementedError('(1,2,2) transposed convolution needs D*H*W % 4 == 0')
        else:                              # odd-sized tensors: vector-ALU kernel
            _lib.check(self.lib.wmh_convtranspose3d_k2s2_fwd(
                _lib.ptr(x), _lib.ptr(ss), _lib.ptr(w), _lib.ptr(y), N, Cin, cout, D, H, W, LRELU_SLOPE,
                self._stream()), 'wmh_convtranspose3d_k2s2_fwd')
        return y

    # ---- channels-last bf16 levels (storage='bf16') ---------------------------------------------
    def _setup_channels_last(self, sd):
        """Decide which resolution levels run channels-last (rows of the level that are multiples of 32 voxels, isotropic
        3x3x3 / 2x2x2 plan there) and pack the weights of their layers.  Fails at construction when level 0 cannot."""
        plan = self.plan
        W = int(plan.patch_size[2])
        nlev = plan.num_pool + 1
        ncl = 0
        for lv in range(nlev):
            iso = list(plan.conv_kernel_sizes[lv]) == [3, 3, 3] and (lv == 0 or list(plan.pool_op_kernel_sizes[lv - 1]) == [2, 2, 2])
            wl = W >> lv
            if not iso or wl < 32 or wl % 32 or (W % (1 << lv)):
                break
            ncl = lv + 1
        problems = []
        if ncl == 0:
            problems.append('the last patch axis must be a multiple of 32 voxels and the first stage isotropic (3x3x3)')
        if plan.num_input_channels != 1:
            problems.append('one input modality (has %d)' % plan.num_input_channels)
        feats = plan.features()
        if any(f % 32 for f in feats[:max(ncl, 1)]):
            problems.append('feature counts of the channels-last levels must be multiples of 32 (%s)' % feats[:ncl])
        if plan.num_classes != 2 or feats[0] != 32:
            problems.append('segmentation head kernel: 32 feature maps at full resolution and 2 classes')
        if problems:
            raise ValueError("storage='bf16' is not available for this plan: " + '; '.join(problems))
        self.ncl = ncl
        lib, dev = self.lib, self.device
        e0 = self.enc[0]
        wf = np.empty(lib.wmh_cl_conv3_first_packed_halfs(e0.cout), dtype=np.uint16)
        _lib.check(lib.wmh_cl_conv3_first_pack_weights(e0.w_host.ctypes.data_as(C.c_void_p), e0.cout, wf.ctypes.data_as(C.c_void_p)),
                   'wmh_cl_conv3_first_pack_weights')
        self._cl_first_wf = torch.from_numpy(wf.view(np.int16)).to(dev)
        nd = len(self.dec)
        for u, st in enumerate(self.dec):
            level = nd - 1 - u                     # level this stage up-samples TO
            if level < ncl and level + 1 < ncl:    # both sides channels-last: the cl transposed convolution
                tu_host = _np(sd['tu.%d.weight' % u])
                wt = np.empty(lib.wmh_cl_convt_packed_halfs(st['cdown'], st['cskip']), dtype=np.uint16)
                _lib.check(lib.wmh_cl_convt_pack_weights(tu_host.ctypes.data_as(C.c_void_p), st['cdown'], st['cskip'],
                                                         wt.ctypes.data_as(C.c_void_p)), 'wmh_cl_convt_pack_weights')
                st['tu_wf_cl'] = torch.from_numpy(wt.view(np.int16)).to(dev)

    def _account(self, flop, nbytes):
        if self.acct is not None:
            self.acct['flop'] += flop
            self.acct['bytes'] += nbytes

    def _finalize(self, layer, part, npart, ss, N):
        _lib.check(self.lib.wmh_instnorm_finalize(_lib.ptr(part), npart, _lib.ptr(layer.gamma), _lib.ptr(layer.beta), IN_EPS,
                                                  _lib.ptr(ss), None, N, layer.cout, self._stream()), 'wmh_instnorm_finalize')

    def _cl_conv(self, key, layer, x0, ss0, x1=None, ss1=None):
        """x0 (x1): CL16 tensors [N, C / 8, D, H, W, 8] (torch.bfloat16).  Stride 1 (one or two sources) or stride 2 (one source)."""
        N, C0, D, H, W = self._cl_dims(x0)
        C1 = 0 if x1 is None else self._cl_dims(x1)[1]
        assert C0 + C1 == layer.cin and layer.kz == 3
        lib = self.lib
        s = layer.stride
        Do, Ho, Wo = (D - 1) // s + 1, (H - 1) // s + 1, (W - 1) // s + 1
        if s == 1:
            npart = _lib.check(lib.wmh_cl_conv3_num_partials(D, H, W), 'wmh_cl_conv3_num_partials')
        else:
            assert x1 is None and layer.sz == 2
            npart = _lib.check(lib.wmh_cl_conv3s2_num_partials(Do, Ho, Wo), 'wmh_cl_conv3s2_num_partials')
        y = self._buf(key + '.ycl', self._cl_shape(N, layer.cout, Do, Ho, Wo), torch.bfloat16)
        part = self._buf(key + '.part', (N, layer.cout, npart, 4))
        ss = self._buf(key + '.ss', (N, layer.cout, 2))
        wf = layer.packed_bf16(0)
        timed = 0 in self.timed_cfgs and s == 1
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if s == 1:
            _lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wf),
                                            _lib.ptr(layer.bias), _lib.ptr(y), _lib.ptr(part), N, D, H, W, layer.cout, LRELU_SLOPE,
                                            self._stream()), 'wmh_cl_conv3_fwd')
        else:
            _lib.check(lib.wmh_cl_conv3s2_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(wf), _lib.ptr(layer.bias), _lib.ptr(y),
                                              _lib.ptr(part), N, D, H, W, layer.cout, LRELU_SLOPE, self._stream()), 'wmh_cl_conv3s2_fwd')
        vox = N * Do * Ho * Wo
        flop = 2.0 * layer.cin * layer.cout * 27 * vox
        nbytes = 2.0 * N * layer.cin * D * H * W + 2.0 * layer.cout * vox + 2.0 * layer.cout * layer.cin * 27
        if timed:
            e1.record()
            self.timings.append((0, e0, e1, flop, nbytes))
        self._account(flop, nbytes)
        self._finalize(layer, part, npart, ss, N)
        return y, ss

    @staticmethod
    def _cl_shape(N, C, D, H, W):
        """torch shape of a CL16 tensor (cl_common.h): 8-channel planes, [n][c / 8][z][y][x][c % 8]"""
        assert C % 8 == 0
        return (N, C // 8, D, H, W, 8)

    @staticmethod
    def _cl_dims(t):
        """(N, C, D, H, W) of a CL16 tensor"""
        N, C8, D, H, W, e = t.shape
        assert e == 8
        return N, 8 * C8, D, H, W

    def _to_cl(self, key, x):
        """NCDHW fp32 -> CL16 (level boundary; raw values, the pending scale/shift stays valid)."""
        N, Cc, D, H, W = x.shape
        y = self._buf(key + '.cl', self._cl_shape(N, Cc, D, H, W), torch.bfloat16)
        _lib.check(self.lib.wmh_cl_from_ncdhw(_lib.ptr(x), _lib.ptr(y), N, Cc, D * H * W, self._stream()), 'wmh_cl_from_ncdhw')
        self._account(0.0, 6.0 * x.numel())
        return y

    def _from_cl(self, key, x):
        N, Cc, D, H, W = self._cl_dims(x)
        y = self._buf(key + '.nc', (N, Cc, D, H, W))
        _lib.check(self.lib.wmh_cl_to_ncdhw(_lib.ptr(x), _lib.ptr(y), N, Cc, D * H * W, self._stream()), 'wmh_cl_to_ncdhw')
        self._account(0.0, 6.0 * x.numel())
        return y

    def _forward_features_cl(self, x, deep_supervision=False):
        """The forward graph with the first `ncl` resolution levels channels-last (see _setup_channels_last)."""
        if deep_supervision:
            raise NotImplementedError("storage='bf16' serves inference (the last segmentation head only)")
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device == self.device
        lib, ncl = self.lib, self.ncl
        N, _, D, H, W = x.shape
        nenc = len(self.enc)
        skips = []
        cur = cur_ss = None
        for i, layer in enumerate(self.enc):
            level = i // 2
            if i == 0:
                npart = _lib.check(lib.wmh_cl_conv3_first_num_partials(D, H, W), 'wmh_cl_conv3_first_num_partials')
                cur = self._buf('enc0.ycl', self._cl_shape(N, layer.cout, D, H, W), torch.bfloat16)
                part = self._buf('enc0.part', (N, layer.cout, npart, 4))
                cur_ss = self._buf('enc0.ss', (N, layer.cout, 2))
                _lib.check(lib.wmh_cl_conv3_first_fwd(_lib.ptr(x), _lib.ptr(self._cl_first_wf), _lib.ptr(layer.bias), _lib.ptr(cur),
                                                      _lib.ptr(part), N, D, H, W, layer.cout, self._stream()), 'wmh_cl_conv3_first_fwd')
                self._account(2.0 * 27 * layer.cout * x.numel(), 4.0 * x.numel() + 2.0 * layer.cout * x.numel())
                self._finalize(layer, part, npart, cur_ss, N)
            elif level < ncl:
                cur, cur_ss = self._cl_conv('enc%d' % i, layer, cur, cur_ss)
            else:
                if cur.dtype == torch.bfloat16:            # first layer below the channels-last levels
                    cur = self._from_cl('enc%d.in' % i, cur)
                cur, cur_ss = self._conv('enc%d' % i, layer, cur, cur_ss)
            if i % 2 == 1 and i < nenc - 2:
                skips.append((cur, cur_ss))
        nd = len(self.dec)
        outs = []
        for u, st in enumerate(self.dec):
            level = nd - 1 - u
            skip, skip_ss = skips[-(u + 1)]
            if level < ncl:
                if cur.dtype == torch.bfloat16:
                    Nn, Cc, Dd, Hh, Ww = self._cl_dims(cur)
                    up = self._buf('tu%d.ycl' % u, self._cl_shape(Nn, st['cskip'], 2 * Dd, 2 * Hh, 2 * Ww), torch.bfloat16)
                    _lib.check(lib.wmh_cl_convt_fwd(_lib.ptr(cur), _lib.ptr(cur_ss), _lib.ptr(st['tu_wf_cl']), _lib.ptr(up), Nn, Cc,
                                                    st['cskip'], Dd, Hh, Ww, LRELU_SLOPE, self._stream()), 'wmh_cl_convt_fwd')
                    self._account(2.0 * Cc * st['cskip'] * 8 * Nn * Dd * Hh * Ww, 2.0 * cur.numel() + 2.0 * up.numel() + 2.0 * Cc * st['cskip'] * 8)
                else:                                       # up-sampling out of the NCDHW levels
                    up = self._to_cl('tu%d' % u, self._convT('tu%d' % u, st['tu_w'], st['tu_wf'], cur, cur_ss, st['cskip'], st['kt']))
                cur, cur_ss = self._cl_conv('dec%d.0' % u, st['convs'][0], up, None, skip, skip_ss)
                cur, cur_ss = self._cl_conv('dec%d.1' % u, st['convs'][1], cur, cur_ss)
            else:
                up = self._convT('tu%d' % u, st['tu_w'], st['tu_wf'], cur, cur_ss, st['cskip'], st['kt'], wf_x3=st['tu_wf_x3'])
                cur, cur_ss = self._conv('dec%d.0' % u, st['convs'][0], up, None, skip, skip_ss)
                cur, cur_ss = self._conv('dec%d.1' % u, st['convs'][1], cur, cur_ss)
            if u == nd - 1:
                outs.append((cur, cur_ss, st['seg_w']))
        return outs

    # ---- graph ---------------------------------------------------------------------------------
    def forward_features(self, x, deep_supervision=False):
        """x: [N, Cin, P0, P1, P2] fp32 on the device.  Returns a list of (y, ss, seg_w) per decoder
        stage output that was requested (last = full resolution): y is the raw conv output whose
        InstanceNorm+LeakyReLU is still pending as ss (a torch.bfloat16 tensor at full resolution with storage='bf16')."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device == self.device
        cur, cur_ss = x, None
        skips = []
        nenc = len(self.enc)
        if self.ncl:
            return self._forward_features_cl(x, deep_supervision)
        st16 = self.storage == 'bf16_ncdhw'        # the two encoder and two decoder convolutions + the up-sampling of level 0
        for i, layer in enumerate(self.enc):
            cur, cur_ss = self._conv('enc%d' % i, layer, cur, cur_ss, out16=st16 and i < 2)
            if i % 2 == 1 and i < nenc - 2:
                skips.append((cur, cur_ss))
        outs = []
        for u, st in enumerate(self.dec):
            o16 = st16 and u == len(self.dec) - 1
            skip, skip_ss = skips[-(u + 1)]
            fused = None if o16 else self._dec_first_fused('dec%d.0' % u, st, cur, cur_ss, skip, skip_ss)
            if fused is not None:
                cur, cur_ss = fused
            else:
                up = self._convT('tu%d' % u, st['tu_w'], st['tu_wf'], cur, cur_ss, st['cskip'], st['kt'], out16=o16, wf_x3=st['tu_wf_x3'])
                cur, cur_ss = self._conv('dec%d.0' % u, st['convs'][0], up, None, skip, skip_ss, out16=o16)
            cur, cur_ss = self._conv('dec%d.1' % u, st['convs'][1], cur, cur_ss, out16=o16)
            if deep_supervision or u == len(self.dec) - 1:
                outs.append((cur, cur_ss, st['seg_w']))
        return outs

    def forward_logits(self, x, deep_supervision=False):
        """Logits like Generic_UNet.forward: [N,K,...] (or the deep-supervision tuple, full res first)."""
        outs = self.forward_features(x, deep_supervision)
        K = self.plan.num_classes
        res = []
        for (y, ss, seg_w) in outs:
            if self.ncl:                               # channels-last feature map: through the NCDHW head kernel
                y = self._from_cl('head.in', y)
            N, Cc = y.shape[:2]
            sp = int(np.prod(y.shape[2:]))
            lg = torch.empty((N, K) + tuple(y.shape[2:]), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.wmh_seghead_logits_st(_lib.ptr(y), _lib.ptr(ss), _lib.ptr(seg_w), K, _lib.ptr(lg),
                                                      N, Cc, sp, LRELU_SLOPE, int(y.dtype == torch.bfloat16), self._stream()),
                       'wmh_seghead_logits_st')
            res.append(lg)
        return tuple(res[::-1]) if deep_supervision else res[-1]
