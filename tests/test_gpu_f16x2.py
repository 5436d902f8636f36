"""`precision='f16x2'`: the 3x3x3 convolutions form their products from FP16 pairs (x = hi + lo, 21-22 significant bits per
operand, the partial products on v_mfma_f32_32x32x16_f16 -- all four in the one-role, strided and transposed kernels, three in the
wave-specialised stride-1 kernels (end of round 3) and in the pair weight gradient (round 4): lo x lo, 2^-22 of a product, is left out --, fp32 accumulate, weights
scaled by 2^8 before the split).  Round 2
used bf16 pairs (16 bits: 3.4e-6 per layer, 6.1e-5 on the full-size softmax); with fp16 pairs a layer's error against float64 is
at or below the exact-fp32 kernel's.  The gate: softmax within 5e-5 of the oracle (the fp32 path's own bar is 1e-4), identical
argmax outside the margin band."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from deepwmh_amd import _lib
    return _lib.load()


CASES = [  # N, D, H, W, C0, C1, Cout, pending norm on source 0 / 1 [, stride]
    (2, 8, 8, 32, 32, 0, 32, True, False),
    (2, 8, 8, 8, 48, 16, 64, True, True, 1),      # rows of 8 voxels (configuration 24: wave-specialised, split-K slices; and configuration 4, one-role kernel)
    (3, 8, 16, 8, 40, 0, 96, True, False, 1),     # ... several tiles per sample, a channel-group tail
    (1, 5, 9, 7, 24, 0, 40, True, False, 1),      # ... ragged
    (2, 16, 16, 16, 32, 0, 64, True, False, 2),   # stride 2, output rows of 8 (configuration 5)
    (2, 8, 32, 32, 32, 0, 128, True, False, 2),   # stride 2, output rows of 16 (configuration 3)
    (1, 7, 20, 27, 16, 0, 64, False, False, 2),   # ... odd extents
    (2, 8, 16, 64, 32, 0, 64, True, False, 2),    # stride 2, output rows of 32 (configuration 1: the first convolution of the 64^3 / 32^3 stages)
    (1, 7, 9, 52, 16, 0, 40, True, False, 2),     # ... ragged tiles, channel counts that are not multiples of the chunk / the tile (rows: multiples of 4 voxels)
    (1, 6, 6, 128, 8, 0, 128, False, False, 2),   # ... two tiles per row, no pending norm
    (3, 9, 21, 72, 20, 0, 72, True, False, 2),    # ... a channel count that is no multiple of the chunk, two output-channel groups with a tail, many tiles per workgroup
    (1, 5, 9, 36, 12, 0, 40, True, False),        # channel counts that are not multiples of the chunk / the tile; ragged tiles
    (1, 6, 10, 32, 16, 24, 32, False, True),      # two sources (the decoder's concatenation), the second one normalised
    (2, 4, 12, 64, 64, 64, 64, True, True),       # the 64-channel tile (configuration 21)
    (2, 18, 6, 16, 48, 0, 64, True, False),       # rows of 16 voxels (configuration 2)
    (1, 5, 9, 8, 24, 0, 40, True, False, 1),      # configuration 24 on ragged tiles (5 x 9 rows of 8), channel counts that are no multiples of the chunk / the tile
    (1, 7, 18, 28, 16, 0, 72, False, False, 2),   # configuration 25 (stride 2, output 4 x 9 x 14): ragged in every direction, a channel-group tail
    (2, 9, 14, 12, 24, 0, 40, True, False, 2),    # configuration 26 (stride 2, output 5 x 7 x 6)
    (2, 16, 16, 16, 256, 0, 256, True, False),    # a 16^3 level at the training batch: 64 tiles, so the wave-specialised kernel runs four slices of the channel chunks per tile
    (2, 8, 16, 32, 40, 24, 64, True, True),       # ... two sources, chunk slices of unequal length
]


@pytest.mark.parametrize('slabs', [True, False], ids=['split-K where the library picks it', 'no slab memory'])
@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_f16x2_against_float64(lib, case, slabs):
    from deepwmh_amd import _lib
    P = _lib.ptr
    N, D, H, W, C0, C1, Cout, has0, has1 = case[:9]
    stride = case[9] if len(case) > 9 else 1
    g = torch.Generator().manual_seed(N * 1000 + W + C0)
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W, generator=g)
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    mk = lambda c: torch.stack([1 + 0.3 * torch.randn(N, c, generator=g), 0.2 * torch.randn(N, c, generator=g)], -1).contiguous()
    ss0, ss1 = (mk(C0) if has0 else None), (mk(C1) if (C1 and has1) else None)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(27 * Cin)
    bias = 0.1 * torch.randn(Cout, generator=g)

    def act(x, ss):
        x = x.double()
        return x if ss is None else torch.nn.functional.leaky_relu(x * ss[..., 0, None, None, None].double() + ss[..., 1, None, None, None].double(), 0.01)
    xin = act(x0, ss0) if x1 is None else torch.cat([act(x0, ss0), act(x1, ss1)], 1)
    ref = torch.nn.functional.conv3d(xin, w.double(), bias.double(), padding=1, stride=stride)
    Do, Ho, Wo = ref.shape[2:]
    cfg0 = _lib.check(lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, stride), 'config')
    # the configuration the engine uses, and -- where that is the wave-specialised one for rows of 8 voxels (24) -- the one-role kernel as well
    for cfg in sorted({lib.wmh_conv3d_k3_f16x2_config(cfg0, Cout, W), lib.wmh_conv3d_k3_bf16_config(cfg0, Cout, W)}):
        assert lib.wmh_conv3d_k3_f16x2_supported(cfg)
        wh = np.ascontiguousarray(w.numpy().reshape(Cout, Cin, 27))
        wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), np.uint16)
        _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, wh.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
        wfd = torch.from_numpy(wf.view(np.int16)).cuda()
        y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
        npart = _lib.check(lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, Do, Ho, Wo), 'npart')
        nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, Cin, N, D, H, W, Cout) if slabs else 0
        ws = torch.full((nws,), float('nan'), device='cuda') if nws else None
        part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
        x0d, x1d = x0.cuda(), (x1.cuda() if C1 else None)
        ss0d, ss1d, bd = (ss0.cuda() if ss0 is not None else None), (ss1.cuda() if ss1 is not None else None), bias.cuda()
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x0d), P(ss0d), C0, P(x1d), P(ss1d), C1, P(wfd), P(bd), P(y), P(part), P(ws), N, D, H, W, Cout,
                                                0.01, None), 'wmh_conv3d_k3_fwd_f16x2')
        torch.cuda.synchronize()
        got = y.cpu().double()
        assert bool(torch.isfinite(got).all())
        rel = float((got - ref).norm() / ref.norm())
        mx = float((got - ref).abs().max()) / float(ref.abs().max())
        # 21-22 significant bits per operand; the exact-fp32 kernel's own figures on these shapes are 0.5-1.4e-6 / 1.6-3.7e-6
        # (measured for this kernel: 3.5e-7 .. 9e-7 / 0.9 .. 2.9e-6; round 2's bf16 pairs: 3.4e-6 / 4e-6)
        assert rel <= 2e-6 and mx <= 6e-6, (rel, mx)
        # InstanceNorm statistics from the fp32 accumulators, through the usual finalisation
        ss = torch.empty(N, Cout, 2, device='cuda')
        _lib.check(lib.wmh_instnorm_finalize(P(part), npart, None, None, 1e-5, P(ss), None, N, Cout, None), 'finalize')
        mean, var = ref.mean((2, 3, 4)), ref.var((2, 3, 4), unbiased=False)
        ss_ref = torch.stack([1 / torch.sqrt(var + 1e-5), -mean / torch.sqrt(var + 1e-5)], -1)
        assert float((ss.cpu().double() - ss_ref).abs().max()) <= 2e-4 * (1 + float(ss_ref.abs().max()))


def test_strided_pair_kernel_refuses_rows_that_are_not_multiples_of_four(lib):
    """Configuration 1 (stride 2, output rows of 32) loads aligned 16-byte quads since round 5: an input row of 50 voxels is refused by the
    library (the engine then runs the exact-fp32 kernel for that layer), not computed wrongly."""
    from deepwmh_amd import _lib
    P = _lib.ptr
    N, D, H, W, Cin, Cout = 1, 6, 6, 50, 8, 32
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 2), Cout, W)
    assert cfg == 1
    x = torch.zeros(N, Cin, D, H, W, device='cuda'); y = torch.zeros(N, Cout, 3, 3, 25, device='cuda')
    wf = torch.zeros(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), dtype=torch.int16, device='cuda')
    assert lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x), None, Cin, None, None, 0, P(wf), None, P(y), None, None, N, D, H, W, Cout, 0.01, None) != 0


@pytest.mark.parametrize('shape', [(2, 32, 64, 8, 8, 32), (1, 24, 40, 5, 6, 16), (2, 64, 128, 4, 8, 8)], ids=str)
def test_convtranspose_data_gradient_on_the_strided_pair_kernels(lib, shape):
    """dX of ConvTranspose3d(k=2, s=2) = a stride-2 3x3x3 convolution of dY with the 2-window embedded at taps 1..2 of the
    3-window (image mode 2 of wmh_conv3d_k3_pack_weights_f16x2_device), on the strided pair kernels; against float64 autograd."""
    from deepwmh_amd import _lib
    P = _lib.ptr
    N, Cs, Ci, D, H, W = shape        # the transposed convolution maps Ci -> Cs channels, (D,H,W) -> twice that
    g = torch.Generator().manual_seed(Cs + W)
    w = torch.randn(Ci, Cs, 2, 2, 2, generator=g) / np.sqrt(8 * Ci)
    dy = torch.randn(N, Cs, 2 * D, 2 * H, 2 * W, generator=g)
    x = torch.zeros(N, Ci, D, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv_transpose3d(x, w.double(), None, stride=2).backward(dy.double())
    ref = x.grad
    cfg0 = _lib.check(lib.wmh_conv3d_k3_config(Cs, Ci, 2 * D, 2 * H, 2 * W, 2), 'config')
    # the configuration the trainer uses (the wave-specialised strided kernels: 1, 25, 26) and the one-role kernels (3, 5) behind them
    for cfg in sorted({lib.wmh_conv3d_k3_f16x2_config(cfg0, Ci, 2 * W), lib.wmh_conv3d_k3_bf16_config(cfg0, Ci, 2 * W)}):
        assert cfg in (1, 3, 5, 25, 26) and lib.wmh_conv3d_k3_f16x2_supported(cfg)
        wd = w.cuda()
        wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cs, Ci), dtype=torch.int16, device='cuda')
        _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(wd), Cs, Ci, 0, Cs, 2, P(wf), None), 'pack')
        dx = torch.full((N, Ci, D, H, W), float('nan'), device='cuda')
        nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, Cs, N, 2 * D, 2 * H, 2 * W, Ci)
        ws = torch.empty(nws, device='cuda') if nws else None
        dyd = dy.cuda()
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(dyd), None, Cs, None, None, 0, P(wf), None, P(dx), None, P(ws), N, 2 * D, 2 * H, 2 * W, Ci,
                                                0.01, None), 'wmh_conv3d_k3_fwd_f16x2')
        torch.cuda.synchronize()
        got = dx.cpu().double()
        assert bool(torch.isfinite(got).all())
        rel = float((got - ref).norm() / ref.norm())
        assert rel <= 2e-6, rel
        # mode 1 (the flipped, transposed image of a stride-1 layer) has no meaning for the interleaved image of configuration 1
        if cfg in (1, 25, 26):
            assert lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(wd), Cs, Ci, 0, Cs, 1, P(wf), None) != 0


def test_cycle_stamps_exist_in_the_diagnostic_library_only_and_leave_the_results_alone(lib):
    """wmh_debug_set_conv_stamps: in libdeepwmh_amd_diag.so (-DWMH_DIAG) the persistent kernels add s_memtime cycle counts per workgroup to
    the caller's buffer (bench.py derives the clock the CUs ran at from them); the product library's kernels hold no stamp code and it does not export the hook
    (include/deepwmh_amd_diag.h).  The convolution's output is bit-identical across the two libraries, with and without stamps."""
    from deepwmh_amd import _lib
    dlib = _lib.load_diag()
    assert lib.wmh_diag_build() == 0 and dlib.wmh_diag_build() == 1
    P = _lib.ptr
    N, C, S = 2, 32, 32
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(N, C, S, S, S, device='cuda', generator=g)
    ss = torch.stack([1 + 0.3 * torch.randn(N, C, device='cuda', generator=g), 0.2 * torch.randn(N, C, device='cuda', generator=g)], -1).contiguous()
    w = (torch.randn(C, C, 27, device='cuda', generator=g) / np.sqrt(27 * C)).contiguous()
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(C, C, S, S, S, 1), C, S)
    assert cfg == 0 and lib.wmh_conv3d_k3_f16x2_supported(cfg)
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, C, C), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), C, C, 0, C, 0, P(wf), None), 'pack')
    part = torch.empty(N, C, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, S, S, S), 4, device='cuda')

    def run(which):
        y = torch.full((N, C, S, S, S), float('nan'), device='cuda')
        rc = which.wmh_conv3d_k3_fwd_f16x2(cfg, P(x), P(ss), C, None, None, 0, P(wf), None, P(y), P(part), None, N, S, S, S, C, 0.01, None)
        assert rc == 0
        torch.cuda.synchronize()
        return y
    plain = run(lib)
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
    for name in _lib.DIAG_SIGNATURES:                                # product library: neither stamp code in the kernels nor the hooks (round 6)
        assert not hasattr(lib, name), name
    assert torch.equal(plain, run(dlib))                             # same arithmetic in the diagnostic build
    dlib.wmh_debug_set_conv_stamps(C_void(stamps))
    try:
        stamped = run(dlib)
    finally:
        dlib.wmh_debug_set_conv_stamps(None)
    assert torch.equal(plain, stamped)
    st = stamps.view(-1, 8).cpu().numpy()
    used = st[st[:, 3] > 0]
    assert len(used) > 0 and (used[:, 0] > 0).all()                  # MFMA-phase cycles of every workgroup that had chunks to do
    assert ((used[:, 0] + used[:, 1] + used[:, 2]) < 2e9).all()      # a 32^3 layer takes well under a second of cycles
    stamps.zero_()
    again = run(dlib)
    assert torch.equal(plain, again) and int(stamps.abs().sum()) == 0


def C_void(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def test_engine_f16x2_meets_the_fp32_gate(tiny_net, tiny_io):
    """Tiny net, tiled + mirrored prediction: the same bounds tests/test_gpu_network.py puts on the fp32 path."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    plan, net, sd = tiny_net
    eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2')
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=True).predict(tiny_io['volume'])
    ref = tiny_io['prob_tta']
    d = float(np.abs(prob - ref).max())
    margin = np.abs(ref[1] - ref[0])
    mism = int(((seg != tiny_io['seg_tta']) & (margin > 2e-4)).sum())
    print('f16x2 tiny net: max |softmax - golden| %.3g, argmax mismatches outside the margin band %d' % (d, mism))
    assert d <= 5e-5 and mism == 0
    with pytest.raises(ValueError):
        UNetEngine(plan, sd, 'cuda:0', precision='f16x2', storage='bf16')


def test_engine_f16x2_full_width_patch_vs_fp32_path():
    """BASELINE plan (31 M parameters), a batch of 128^3 patches: against the exact-fp32 HIP path (itself within 1.5e-5 of the oracle)."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    vol = synthetic_flair((128, 128, 160), seed=3)
    x = torch.from_numpy(np.ascontiguousarray(np.stack([vol[:, :, :, :128], vol[:, :, :, 32:]]))).cuda()
    ref = torch.softmax(UNetEngine(plan, sd, 'cuda:0').forward_logits(x), 1)
    eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2')
    a = torch.softmax(eng.forward_logits(x), 1)
    assert torch.equal(a, torch.softmax(eng.forward_logits(x), 1))          # bit-reproducible
    d = float((a - ref).abs().max())
    margin = (ref[:, 1] - ref[:, 0]).abs()
    mism = int((((a[:, 1] > 0.5) != (ref[:, 1] > 0.5)) & (margin > 2e-4)).sum())
    print('f16x2 full width: max |softmax - fp32 path| %.3g, class changes outside the margin band %d' % (d, mism))
    assert d <= 5e-5 and mism == 0


def test_activation_range_of_the_fp16_pair(lib):
    """fp16's range.  The high part is rounded toward zero, so it saturates at 65504 and the low part carries the rest: activations
    up to 131008 in magnitude stay finite (with fp16's coarse spacing up there); beyond that the low part overflows and the output
    is non-finite -- loud, like nnU-Net's own fp16 autocast at 65504, not silently wrong.  (Behind InstanceNorm + LeakyReLU an
    activation is bounded by sqrt(voxels) ~ 1.4e3 times the affine scale.)"""
    from deepwmh_amd import _lib
    P = _lib.ptr
    N, Cin, Cout, D, H, W = 1, 8, 32, 4, 8, 32
    w = np.zeros((Cout, Cin, 27), np.float32)
    w[:, :, 13] = 1.0 / Cin
    cfg = lib.wmh_conv3d_k3_bf16_config(_lib.check(lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1), 'config'), Cout, W)
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), np.uint16)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    for big, finite in ((1.0e5, True), (-1.0e5, True), (3.0e5, False)):
        x = torch.ones(N, Cin, D, H, W)
        x[0, 0, 1, 2, 3] = big
        y = torch.empty(N, Cout, D, H, W, device='cuda')
        xd = x.cuda()
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(xd), None, Cin, None, None, 0, P(wfd), None, P(y), None, None, N, D, H, W, Cout, 0.01, None), 'fwd')
        torch.cuda.synchronize()
        got = y.cpu()
        assert float(got[0, 0, 0, 0, 0]) == pytest.approx(1.0, abs=1e-6)
        if finite:
            assert bool(torch.isfinite(got).all())
            assert float(got[0, 0, 1, 2, 3]) == pytest.approx(big / Cin + 7.0 / 8, rel=2e-4)
        else:
            assert not bool(torch.isfinite(got[0, :, 1, 2, 3]).all())


@pytest.mark.parametrize('C,S', [(320, 8), (256, 16)])
def test_split_k_makes_small_levels_depend_on_the_batch_only_in_the_last_bits(lib, C, S):
    """the split-K factor of the wave-specialised pair kernels follows the launch's tile count, which includes N -- a patch in a
    batch of 1 and the same patch in a batch of 16 are summed in different orders at the 16^3 / 8^3 levels (deterministic for a fixed shape,
    every order a fixed tree).  Pinned here: the two differ by rounding only (<= 1e-6 relative L2, the kernels' own distance from float64
    is 3e-7), and the full batch is item-independent bit for bit."""
    from deepwmh_amd import _lib
    P = _lib.ptr
    g = torch.Generator(device='cuda').manual_seed(11)
    N = 16
    x = torch.randn(N, C, S, S, S, device='cuda', generator=g)
    x[1] = x[0]
    ss = torch.stack([1 + 0.3 * torch.randn(N, C, device='cuda', generator=g), 0.2 * torch.randn(N, C, device='cuda', generator=g)], -1).contiguous()
    ss[1] = ss[0]
    w = (torch.randn(C, C, 27, device='cuda', generator=g) / np.sqrt(27 * C)).contiguous()
    cfg = lib.wmh_conv3d_k3_f16x2_config(lib.wmh_conv3d_k3_config(C, C, S, S, S, 1), C, S)
    assert lib.wmh_conv3d_k3_f16x2_supported(cfg)
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, C, C), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), C, C, 0, C, 0, P(wf), None), 'pack')

    def run(n):
        y = torch.full((n, C, S, S, S), float('nan'), device='cuda')
        nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, C, n, S, S, S, C)
        ws = torch.empty(max(nws, 1), device='cuda')
        part = torch.empty(n, C, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, n, S, S, S), 4, device='cuda')
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x[:n].contiguous()), P(ss[:n].contiguous()), C, None, None, 0, P(wf), None, P(y), P(part),
                                               P(ws) if nws else None, n, S, S, S, C, 0.01, None), 'fwd')
        torch.cuda.synchronize()
        return y
    y16, y1 = run(16), run(1)
    assert torch.equal(y16[0], y16[1])                               # same patch, two slots of one batch: identical bits
    rel = float((y16[0] - y1[0]).double().norm() / y1[0].double().norm())
    assert rel <= 1e-6, rel


@pytest.mark.parametrize('cfg,Cin,Cout', [(0, 32, 32), (0, 20, 40), (21, 64, 64), (2, 256, 256), (24, 320, 320), (1, 32, 64), (25, 128, 256), (26, 256, 320)])
def test_device_packer_builds_the_host_packers_image(lib, cfg, Cin, Cout):
    """UNetEngine builds its f16x2 weight images with wmh_conv3d_k3_pack_weights_f16x2_device since round 6 (the host loops were most of a fresh
    process's first forward): every configuration's image must be the host packer's, bit for bit -- ragged channel counts and weights whose low parts
    are subnormal included."""
    import ctypes
    from deepwmh_amd import _lib
    P = _lib.ptr
    if not lib.wmh_conv3d_k3_f16x2_supported(cfg):
        pytest.skip('configuration %d has no f16x2 kernel' % cfg)
    rng = np.random.default_rng(cfg + Cin)
    w = (rng.standard_normal((Cout, Cin, 27)) * np.exp(rng.uniform(-9, 3, (Cout, Cin, 27)))).astype(np.float32)
    w = np.clip(w, -200, 200)
    n = lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout)
    host = np.zeros(n, np.uint16)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(ctypes.c_void_p), Cin, Cout, host.ctypes.data_as(ctypes.c_void_p)), 'host pack')
    wd = torch.from_numpy(w).cuda()
    dev = torch.full((n,), 0x7e00, dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(wd), Cin, Cout, 0, Cin, 0, P(dev), None), 'device pack')
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint16), host)
