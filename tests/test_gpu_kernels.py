"""GPU parity tests, kernel by kernel, through the C ABI (ctypes) against torch-CPU fp32 references
of the same op (the arithmetic primitives nnU-Net itself is made of)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SLOPE = 0.01


@pytest.fixture(scope='module')
def lib():
    from deepwmh_amd import _lib
    lib = _lib.load()
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    _lib.check(lib.wmh_device_check(0), 'wmh_device_check')
    return lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def P(t):
    from deepwmh_amd import _lib
    return _lib.ptr(t)


def iarr(v):
    from deepwmh_amd import _lib
    return _lib.iarr(v)


def test_mfma_fragment_layout(lib):
    # exact small integers, asymmetric B: a transposed or mis-mapped fragment cannot pass
    rng = np.random.default_rng(0)
    a = rng.integers(-8, 9, (32, 2)).astype(np.float32)
    b = rng.integers(-8, 9, (2, 32)).astype(np.float32)
    b[0, :] += np.arange(32, dtype=np.float32)
    d = torch.full((32, 32), float('nan'), device='cuda')
    ad, bd = dev(a), dev(b)
    assert lib.wmh_selftest_mfma_32x32x2(P(ad), P(bd), P(d), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), a @ b)


def _lrelu_affine(x, ss):
    if ss is None:
        return x
    y = x * ss[..., 0][:, :, None, None, None] + ss[..., 1][:, :, None, None, None]
    return F.leaky_relu(y, SLOPE)


CONV_CASES = [
    # N, C0, C1, Cout, D, H, W, stride, ss0, ss1, expected cfg
    (1, 1, 0, 8, 32, 32, 32, 1, False, False, 8),
    (2, 2, 0, 40, 12, 16, 40, 1, True, False, 8),
    (2, 8, 0, 8, 16, 24, 40, 1, True, False, 0),
    (1, 32, 0, 32, 8, 16, 64, 1, True, False, 0),
    (1, 6, 0, 40, 9, 17, 33, 1, False, False, 0),
    (2, 8, 8, 8, 16, 16, 32, 1, False, True, 0),
    (1, 4, 12, 16, 8, 8, 32, 1, True, True, 0),
    (1, 8, 0, 16, 32, 32, 64, 2, True, False, 1),
    (1, 5, 0, 8, 9, 11, 71, 2, False, False, 1),
    (2, 16, 0, 16, 16, 16, 16, 1, True, False, 2),
    (1, 8, 8, 24, 6, 20, 13, 1, False, True, 2),
    (1, 16, 0, 32, 32, 32, 32, 2, True, False, 3),
    (3, 32, 0, 32, 8, 8, 8, 1, True, False, 4),
    (1, 16, 16, 40, 5, 9, 7, 1, True, True, 4),
    (1, 16, 0, 32, 16, 16, 16, 2, True, False, 5),
    (5, 32, 0, 32, 4, 4, 4, 1, True, False, 6),
    (2, 16, 16, 8, 3, 4, 5, 1, False, True, 6),
    (5, 16, 0, 32, 8, 8, 8, 2, True, False, 7),
    (1, 64, 0, 64, 8, 8, 32, 1, True, False, 0),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3d_k3_and_instnorm_partials(lib, case):
    from deepwmh_amd import _lib
    N, C0, C1, Cout, D, H, W, s, use_ss0, use_ss1, exp_cfg = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W, generator=g) * 1.5 + 0.3
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    ss0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1) if use_ss0 else None
    ss1 = torch.stack([0.5 + torch.rand(N, C1, generator=g), 0.3 * torch.randn(N, C1, generator=g)], -1) if use_ss1 else None
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)

    xin = _lrelu_affine(x0, ss0)
    if C1:
        xin = torch.cat((xin, _lrelu_affine(x1, ss1)), 1)
    ref = F.conv3d(xin, w, b, stride=s, padding=1)
    Do, Ho, Wo = ref.shape[2:]

    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, s)
    assert cfg == exp_cfg
    nfl = lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout)
    wf = np.empty(nfl, np.float32)
    wn = w.numpy()
    assert lib.wmh_conv3d_k3_pack_weights(cfg, wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    npart = lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo)
    y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    ss_out = torch.full((N, Cout, 2), float('nan'), device='cuda')
    mr = torch.full((N, Cout, 2), float('nan'), device='cuda')
    d = [t.cuda() if t is not None else None for t in (x0, ss0, x1, ss1)]
    wfd, bd, gd, btd = dev(wf), b.cuda(), gamma.cuda(), beta.cuda()
    nws = lib.wmh_conv3d_k3_workspace_floats(cfg, Cin, N, D, H, W, Cout, s)
    ws = torch.full((nws,), float('nan'), device='cuda') if nws else None
    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(wfd), P(bd), P(y), P(part), P(ws),
                                     N, D, H, W, Cout, s, SLOPE, None), 'conv')
    _lib.check(lib.wmh_instnorm_finalize(P(part), npart, P(gd), P(btd), 1e-5, P(ss_out), P(mr), N, Cout, None), 'fin')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any(), 'conv left output voxels unwritten'
    scale = float(ref.abs().max())
    err = float((yh - ref).abs().max())
    assert err <= 1e-5 * scale + 1e-6, 'conv3d max err %.3g (scale %.3g)' % (err, scale)

    # InstanceNorm3d statistics from the epilogue partials vs torch
    mean = ref.mean((2, 3, 4), keepdim=False).double()
    var = ref.double().var((2, 3, 4), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    mrh = mr.cpu().double()
    assert float((mrh[..., 0] - mean).abs().max()) <= 1e-5 * (1 + float(mean.abs().max()))
    assert float(((mrh[..., 1] - rstd) / rstd).abs().max()) <= 1e-5
    sc = gamma.double()[None] * rstd
    sh = beta.double()[None] - mean * sc
    ssh = ss_out.cpu().double()
    assert float(((ssh[..., 0] - sc) / sc.abs().clamp_min(1e-3)).abs().max()) <= 1e-5
    assert float((ssh[..., 1] - sh).abs().max()) <= 1e-5 * (1 + float(sh.abs().max()))
    # and the normalised tensor a consumer would see equals torch's lrelu(instance_norm(conv))
    full = F.leaky_relu(F.instance_norm(ref, weight=gamma, bias=beta, eps=1e-5), SLOPE)
    mine = F.leaky_relu(yh * ss_out.cpu()[..., 0][:, :, None, None, None] + ss_out.cpu()[..., 1][:, :, None, None, None], SLOPE)
    assert float((mine - full).abs().max()) <= 2e-5


ANISO_CASES = [
    # N, Cin, Cout, D, H, W, kz, sz, s
    (2, 8, 8, 6, 16, 40, 1, 1, 1), (1, 16, 24, 5, 20, 13, 1, 1, 1), (3, 32, 32, 4, 8, 8, 1, 1, 1), (5, 32, 32, 2, 4, 4, 1, 1, 1),
    (1, 8, 16, 6, 32, 64, 1, 1, 2), (2, 16, 32, 4, 16, 32, 1, 1, 2), (1, 16, 32, 5, 16, 16, 1, 1, 2), (5, 16, 32, 3, 8, 8, 1, 1, 2),
    (1, 8, 16, 6, 32, 64, 3, 1, 2), (2, 16, 32, 4, 17, 31, 3, 1, 2), (1, 16, 32, 5, 16, 16, 3, 1, 2), (5, 16, 32, 3, 8, 8, 3, 1, 2),
    (1, 1, 8, 4, 32, 32, 1, 1, 1),
]


@pytest.mark.parametrize('case', ANISO_CASES, ids=[str(c) for c in ANISO_CASES])
def test_conv3d_anisotropic_kernels_and_strides(lib, case):
    # nnU-Net's anisotropic plans: kernel (kz,3,3), stride (sz,s,s)
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W, kz, sz, s = case
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1)
    w = torch.randn(Cout, Cin, kz, 3, 3, generator=g) * (2.0 / (9 * kz * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv3d(_lrelu_affine(x, ss), w, b, stride=(sz, s, s), padding=(kz // 2, 1, 1))
    Do, Ho, Wo = ref.shape[2:]
    cfg = lib.wmh_conv3d_config(Cin, Cout, D, H, W, kz, sz, s)
    assert cfg >= 9
    wf = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout), np.float32)
    wn = w.numpy()
    assert lib.wmh_conv3d_k3_pack_weights(cfg, wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    npart = lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo)
    y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    mr = torch.full((N, Cout, 2), float('nan'), device='cuda'); sso = torch.empty(N, Cout, 2, device='cuda')
    nws = lib.wmh_conv3d_k3_workspace_floats(cfg, Cin, N, D, H, W, Cout, s)
    ws = torch.full((nws,), float('nan'), device='cuda') if nws else None
    xd, ssd, wfd, bd = x.cuda(), ss.cuda(), dev(wf), b.cuda()
    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, P(xd), P(ssd), Cin, None, None, 0, P(wfd), P(bd), P(y), P(part), P(ws), N, D, H, W, Cout, s,
                                     SLOPE, None), 'conv')
    _lib.check(lib.wmh_instnorm_finalize(P(part), npart, None, None, 1e-5, P(sso), P(mr), N, Cout, None), 'fin')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    assert float((yh - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
    mean = ref.mean((2, 3, 4)).double()
    assert float((mr.cpu().double()[..., 0] - mean).abs().max()) <= 1e-5 * (1 + float(mean.abs().max()))


@pytest.mark.parametrize('shape', [(2, 16, 8, 3, 4, 4), (1, 24, 20, 5, 8, 16), (2, 64, 32, 2, 8, 8)])
def test_convtranspose3d_anisotropic_1x2x2(lib, shape):
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(15)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 1, 2, 2, generator=g) * 0.2
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1)
    ref = F.conv_transpose3d(_lrelu_affine(x, ss), w, None, stride=(1, 2, 2))
    wf = np.empty(lib.wmh_convtranspose3d_packed_floats(Cin, Cout, 1), np.float32)
    wn = w.numpy()
    assert lib.wmh_convtranspose3d_pack_weights(wn.ctypes.data_as(C.c_void_p), Cin, Cout, 1, wf.ctypes.data_as(C.c_void_p)) == 0
    y = torch.full((N, Cout, D, 2 * H, 2 * W), float('nan'), device='cuda')
    xd, ssd, wd = x.cuda(), ss.cuda(), dev(wf)
    _lib.check(lib.wmh_convtranspose3d_mfma_fwd(P(xd), P(ssd), P(wd), P(y), N, Cin, Cout, D, H, W, 1, SLOPE, None), 'convT 1x2x2')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    assert float((yh - ref).abs().max()) <= 1e-5 * (1 + float(ref.abs().max()))


BF16_CASES = [(1, 32, 0, 32, 8, 16, 64, True, False), (2, 16, 16, 8, 8, 8, 32, False, True), (1, 6, 0, 40, 5, 9, 33, True, False),
              (2, 64, 0, 64, 8, 8, 16, True, False), (1, 16, 32, 24, 6, 20, 13, True, True),
              # 64-channel tiles (configuration 21): whole and ragged tiles, two sources, two cout groups
              (2, 64, 0, 64, 8, 8, 32, True, False), (1, 64, 64, 64, 6, 9, 36, False, True), (1, 24, 0, 128, 5, 7, 44, True, False)]


@pytest.mark.parametrize('case', BF16_CASES, ids=[str(c) for c in BF16_CASES])
def test_conv3d_k3_bf16_operands(lib, case):
    # reference: the same convolution on operands rounded to bf16 (torch's round-to-nearest-even cast), fp32 accumulate
    from deepwmh_amd import _lib
    N, C0, C1, Cout, D, H, W, use_ss0, use_ss1 = case
    g = torch.Generator().manual_seed(97)
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W, generator=g) * 1.5 + 0.3
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    ss0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1) if use_ss0 else None
    ss1 = torch.stack([0.5 + torch.rand(N, C1, generator=g), 0.3 * torch.randn(N, C1, generator=g)], -1) if use_ss1 else None
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    xin = _lrelu_affine(x0, ss0)
    if C1:
        xin = torch.cat((xin, _lrelu_affine(x1, ss1)), 1)
    r = lambda t: t.to(torch.bfloat16).float()
    ref = F.conv3d(r(xin), r(w), b, padding=1)
    full = F.conv3d(xin, w, b, padding=1)
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1)
    assert lib.wmh_conv3d_k3_bf16_supported(cfg)
    cfg = lib.wmh_conv3d_k3_bf16_config(cfg, Cout, W)          # 64-channel tile (configuration 21) where it applies
    assert cfg == (21 if (Cout % 64 == 0 and W % 4 == 0 and W >= 24) else cfg)
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cin, Cout), np.uint16)
    wn = w.numpy()
    assert lib.wmh_conv3d_k3_pack_weights_bf16(cfg, wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, D, H, W)
    y = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    mr = torch.full((N, Cout, 2), float('nan'), device='cuda'); ss_out = torch.empty(N, Cout, 2, device='cuda')
    d = [t.cuda() if t is not None else None for t in (x0, ss0, x1, ss1)]
    wfd, bd = dev(wf.view(np.int16)), b.cuda()
    _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(wfd), P(bd), P(y), P(part),
                                          N, D, H, W, Cout, SLOPE, None), 'conv bf16')
    _lib.check(lib.wmh_instnorm_finalize(P(part), npart, None, None, 1e-5, P(ss_out), P(mr), N, Cout, None), 'fin')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    scale = float(ref.abs().max())
    err = float((yh - ref).abs().max())
    # fmaf(x, scale, shift) vs torch's x*scale+shift can land on the other side of a bf16 rounding boundary for a few
    # inputs (one bf16 ulp = 2^-8 relative on one operand): allow a handful of such flips, nothing larger
    assert err <= 5e-4 * scale + 1e-5, 'vs bf16-rounded reference: %.3g (scale %.3g)' % (err, scale)
    print('bf16 operands vs full fp32: max abs %.3g (scale %.3g)' % (float((yh - full).abs().max()), scale))
    mean = yh.mean((2, 3, 4)).double()
    assert float((mr.cpu().double()[..., 0] - mean).abs().max()) <= 1e-5 * (1 + float(mean.abs().max()))


def test_instnorm_stats_and_apply(lib):
    from deepwmh_amd import _lib
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 7, 9, 10, 11, generator=g) * 3 + 20.0      # large mean: cancellation-prone
    gamma, beta = 1 + 0.2 * torch.randn(7, generator=g), 0.1 * torch.randn(7, generator=g)
    xd = x.cuda()
    ss = torch.empty(3, 7, 2, device='cuda')
    y = torch.empty_like(xd)
    gd, bd = gamma.cuda(), beta.cuda()      # keep device copies alive: a freed temporary's block is reused at once
    _lib.check(lib.wmh_instnorm_stats(P(xd), P(gd), P(bd), 1e-5, P(ss), None, 3, 7, 990, None), 'stats')
    _lib.check(lib.wmh_instnorm_lrelu_apply(P(xd), P(ss), P(y), 3, 7, 990, SLOPE, None), 'apply')
    torch.cuda.synchronize()
    ref = F.leaky_relu(F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5), SLOPE)
    assert float((y.cpu() - ref).abs().max()) <= 3e-5


@pytest.mark.parametrize('shape', [(2, 16, 8, 4, 4, 4, True), (1, 8, 4, 5, 6, 7, False), (2, 12, 20, 3, 16, 33, True)])
def test_convtranspose3d_k2s2(lib, shape):
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W, use_ss = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g) * 0.2
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1) if use_ss else None
    ref = F.conv_transpose3d(_lrelu_affine(x, ss), w, None, stride=2)
    y = torch.full((N, Cout, 2 * D, 2 * H, 2 * W), float('nan'), device='cuda')
    xd, ssd, wd = x.cuda(), (ss.cuda() if use_ss else None), w.cuda()
    _lib.check(lib.wmh_convtranspose3d_k2s2_fwd(P(xd), P(ssd), P(wd), P(y),
                                                N, Cin, Cout, D, H, W, SLOPE, None), 'convT')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    assert float((yh - ref).abs().max()) <= 1e-5 * (1 + float(ref.abs().max()))


@pytest.mark.parametrize('shape', [(2, 16, 8, 4, 4, 4, True), (1, 8, 4, 2, 6, 8, False), (2, 24, 20, 3, 16, 32, True),
                                   (3, 64, 32, 8, 8, 8, True), (1, 40, 72, 4, 4, 5, True)])
def test_convtranspose3d_k2s2_mfma(lib, shape):
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W, use_ss = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, 2, generator=g) * 0.2
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1) if use_ss else None
    ref = F.conv_transpose3d(_lrelu_affine(x, ss), w, None, stride=2)
    wf = np.empty(lib.wmh_convtranspose3d_k2s2_packed_floats(Cin, Cout), np.float32)
    wn = w.numpy()
    assert lib.wmh_convtranspose3d_k2s2_pack_weights(wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    y = torch.full((N, Cout, 2 * D, 2 * H, 2 * W), float('nan'), device='cuda')
    xd, ssd, wd = x.cuda(), (ss.cuda() if use_ss else None), dev(wf)
    _lib.check(lib.wmh_convtranspose3d_k2s2_mfma_fwd(P(xd), P(ssd), P(wd), P(y), N, Cin, Cout, D, H, W, SLOPE, None), 'convT mfma')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    assert float((yh - ref).abs().max()) <= 1e-5 * (1 + float(ref.abs().max()))


@pytest.mark.parametrize('shape', [(2, 16, 8, 4, 4, 4, True, 2), (1, 8, 4, 2, 6, 8, False, 2), (2, 24, 20, 3, 16, 32, True, 2), (3, 64, 32, 8, 8, 8, True, 2),
                                   (1, 40, 72, 4, 4, 5, True, 2), (1, 33, 9, 3, 5, 7, True, 2), (2, 64, 32, 4, 8, 16, True, 1)])
def test_convtranspose3d_f16x2_pairs(lib, shape):
    """Transposed convolution from fp16 pairs against float64: fp32-grade (the exact-fp32 MFMA kernel's own distance is printed), channel
    tails (Cin not a multiple of 8 / 32), odd voxel counts, the (1,2,2) form, weights over four decades."""
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W, use_ss, kt = shape
    g = torch.Generator().manual_seed(14)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    w = torch.randn(Cin, Cout, kt, 2, 2, generator=g) * torch.exp(torch.empty(Cin, Cout, 1, 1, 1).uniform_(-7, 1, generator=g))
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1) if use_ss else None
    act = _lrelu_affine(x, ss)                                   # fp32, as the kernel forms it
    ref = F.conv_transpose3d(act.double(), w.double(), None, stride=(kt, 2, 2))
    xd, ssd, wd = x.cuda(), (ss.cuda() if use_ss else None), w.cuda()
    wf = torch.zeros(lib.wmh_convtranspose3d_packed_dwords_f16x2(Cin, Cout, kt), dtype=torch.int32, device='cuda')
    _lib.check(lib.wmh_convtranspose3d_pack_weights_f16x2_device(P(wd), Cin, Cout, kt, P(wf), None), 'convT pack f16x2')
    y = torch.full((N, Cout, kt * D, 2 * H, 2 * W), float('nan'), device='cuda')
    _lib.check(lib.wmh_convtranspose3d_fwd_f16x2(P(xd), P(ssd), P(wf), P(y), N, Cin, Cout, D, H, W, kt, SLOPE, None), 'convT f16x2')
    torch.cuda.synchronize()
    yh = y.cpu().double()
    assert not torch.isnan(yh).any()
    scale = float(ref.abs().max())
    e16 = float((yh - ref).abs().max()) / scale
    r16 = float((yh - ref).norm() / ref.norm())
    line = 'f16x2 max err / max|ref| %.2e, rel. L2 %.2e' % (e16, r16)
    if (D * H * W) % 4 == 0:
        wf32 = torch.empty(lib.wmh_convtranspose3d_packed_floats(Cin, Cout, kt), device='cuda')
        _lib.check(lib.wmh_convtranspose3d_pack_weights_device(P(wd), Cin, Cout, kt, P(wf32), None), 'pack')
        y32 = torch.empty_like(y)
        _lib.check(lib.wmh_convtranspose3d_mfma_fwd(P(xd), P(ssd), P(wf32), P(y32), N, Cin, Cout, D, H, W, kt, SLOPE, None), 'convT mfma')
        torch.cuda.synchronize()
        line += '  (exact-fp32 kernel: %.2e / %.2e)' % (float((y32.cpu().double() - ref).abs().max()) / scale, float((y32.cpu().double() - ref).norm() / ref.norm()))
    print(line)
    assert e16 <= 2e-6 and r16 <= 1e-6, line


def test_extract_patches_flips_and_padding(lib):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((2, 10, 20, 13)).astype(np.float32)
    Pz = (12, 8, 16)                         # first and last axes larger than the volume -> centred zero pad
    A = (12, 20, 16)
    pad_lo = [(A[i] - vol.shape[1 + i]) // 2 for i in range(3)]
    padded = np.zeros((2,) + A, np.float32)
    padded[:, pad_lo[0]:pad_lo[0] + 10, pad_lo[1]:pad_lo[1] + 20, pad_lo[2]:pad_lo[2] + 13] = vol
    origins = [(0, 0, 0), (0, 12, 0), (0, 5, 0), (0, 5, 0)]
    flips = [0, 0, 5, 7]
    out = torch.full((4, 2) + Pz, float('nan'), device='cuda')
    vd = dev(vol)
    _lib.check(lib.wmh_extract_patches(P(vd), P(out), 2, 10, 20, 13, iarr(pad_lo), Pz[0], Pz[1], Pz[2], 4,
                                       iarr([v for o in origins for v in o]), iarr(flips), None), 'extract')
    torch.cuda.synchronize()
    for b, (o, f) in enumerate(zip(origins, flips)):
        ref = padded[:, o[0]:o[0] + Pz[0], o[1]:o[1] + Pz[1], o[2]:o[2] + Pz[2]]
        axes = [ax for bit, ax in ((4, 1), (2, 2), (1, 3)) if f & bit]
        if axes:
            ref = np.flip(ref, axes)
        assert np.array_equal(out[b].cpu().numpy(), ref), 'item %d' % b


@pytest.mark.parametrize('shape', [(2, 77, 2, (8, 8, 12)), (1, 256, 2, (16, 16, 16)), (2, 64, 3, (6, 10, 14)), (2, 320, 4, (4, 4, 4))], ids=str)
def test_seghead_logits_channel_sliced_form(lib, shape):
    """The deep-supervision heads of the stages at 64^3 and below run four channel slices per workgroup and sum them through LDS
    (seghead_logits4s_kernel): against the einsum of the activated tensor, channel counts that do not divide by four, a ragged last workgroup,
    2-4 classes; twice the same bits."""
    from deepwmh_amd import _lib
    n, Cc, K, Pz = shape
    g = torch.Generator().manual_seed(Cc)
    x = torch.randn(n, Cc, *Pz, generator=g)
    ss = torch.stack([0.5 + torch.rand(n, Cc, generator=g), 0.3 * torch.randn(n, Cc, generator=g)], -1)
    wseg = torch.randn(K, Cc, generator=g) / np.sqrt(Cc)
    xd, ssd, wd = x.cuda(), ss.cuda(), wseg.cuda()
    outs = []
    for _ in range(2):
        lg = torch.full((n, K) + tuple(Pz), float('nan'), device='cuda')
        _lib.check(lib.wmh_seghead_logits(P(xd), P(ssd), P(wd), K, P(lg), n, Cc, int(np.prod(Pz)), SLOPE, None), 'logits')
        torch.cuda.synchronize()
        outs.append(lg.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = torch.einsum('kc,ncxyz->nkxyz', wseg.double(), _lrelu_affine(x, ss).double())
    assert float((outs[0].double() - ref).abs().max()) <= 2e-6 * (1 + float(ref.abs().max()))


def test_seghead_softmax_mirror_accumulate_and_finalize(lib):
    from deepwmh_amd import _lib
    from deepwmh_amd.sliding_window import get_gaussian
    g = torch.Generator().manual_seed(3)
    n, Cc, K, Pz, A = 8, 6, 2, (8, 6, 10), (12, 6, 15)
    x = torch.randn(n, Cc, *Pz, generator=g)
    ss = torch.stack([0.5 + torch.rand(n, Cc, generator=g), 0.3 * torch.randn(n, Cc, generator=g)], -1)
    wseg = torch.randn(K, Cc, generator=g)
    gauss = torch.from_numpy(get_gaussian(Pz))
    origin = (3, 0, 4)
    agg0 = torch.rand(K, *A, generator=g)
    agg = agg0.clone().cuda()
    flips = list(range(8))
    xd, ssd, wd, gd = x.cuda(), ss.cuda(), wseg.cuda(), gauss.cuda()
    _lib.check(lib.wmh_seghead_softmax_accumulate(P(xd), P(ssd), P(wd), K, P(gd), P(agg), Cc,
                                                  Pz[0], Pz[1], Pz[2], A[0], A[1], A[2], iarr(origin), n, iarr(flips),
                                                  SLOPE, None), 'seghead')
    lg = torch.empty(n, K, *Pz, device='cuda')
    _lib.check(lib.wmh_seghead_logits(P(xd), P(ssd), P(wd), K, P(lg), n, Cc, int(np.prod(Pz)), SLOPE, None), 'logits')
    torch.cuda.synchronize()
    logits = torch.einsum('kc,ncxyz->nkxyz', wseg, _lrelu_affine(x, ss))
    assert float((lg.cpu() - logits).abs().max()) <= 1e-5
    res = torch.zeros(1, K, *Pz)
    for m in range(8):
        ax = tuple(a for bit, a in ((4, 2), (2, 3), (1, 4)) if m & bit)
        pred = torch.softmax(logits[m:m + 1], 1)
        res += 1 / 8 * (torch.flip(pred, ax) if ax else pred)
    res[:, :] *= gauss
    exp = agg0.clone()
    exp[:, 3:11, 0:6, 4:14] += res[0]
    assert float((agg.cpu() - exp).abs().max()) <= 2e-6

    # count map + normalise + argmax with a crop window
    cnt = torch.zeros(*A, device='cuda')
    for o in [(0, 0, 0), (4, 0, 5), (2, 0, 3)]:
        _lib.check(lib.wmh_accumulate_weight(P(gd), P(cnt), Pz[0], Pz[1], Pz[2], A[0], A[1], A[2], iarr(o), None), 'cnt')
    cref = torch.zeros(*A)
    for o in [(0, 0, 0), (4, 0, 5), (2, 0, 3)]:
        cref[o[0]:o[0] + 8, o[1]:o[1] + 6, o[2]:o[2] + 10] += gauss
    torch.cuda.synchronize()
    assert torch.equal(cnt.cpu(), cref)
    V, lo = (9, 6, 12), (1, 0, 2)
    cnt2 = (cref + 0.5).cuda()
    prob = torch.empty(K, *V, device='cuda')
    seg = torch.empty(V, dtype=torch.uint8, device='cuda')
    _lib.check(lib.wmh_finalize_argmax(P(agg), P(cnt2), P(prob), P(seg), K, A[0], A[1], A[2], iarr(lo), V[0], V[1], V[2], None), 'fin')
    torch.cuda.synchronize()
    pref = (agg.cpu() / (cref + 0.5))[:, 1:10, 0:6, 2:14]
    assert torch.equal(prob.cpu(), pref)
    assert torch.equal(seg.cpu(), pref.argmax(0).to(torch.uint8))


def test_wave_specialised_form_is_bit_identical(tmp_path):
    """The persistent loader/consumer form of the conv (the product's choice for 32- and 16-wide rows, stride 1 and 2, given a tile per
    CU) must reproduce the one-role kernel bit for bit (same arithmetic, same k order).  The one-role kernel for these shapes and the
    split-K pin (split-K changes the summation order) are variant switches, which exist in the diagnostic library only
    (csrc/wmh_common.h): product library without switches against libdeepwmh_amd_diag.so with WMH_WS=0, both with split-K pinned off
    where the product would split -- hence WMH_KSPLIT=1 through the diagnostic library on both sides for the reference run, and the
    product run compared on the cases it does not split."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np, torch, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "from deepwmh_amd import _lib\n"
        "lib = _lib.load()\n"
        "g = torch.Generator().manual_seed(3)\n"
        "res_y, res_p = [], []\n"
        "for (N, C0, C1, Cout, D, H, W, st) in [(2, 12, 0, 40, 9, 20, 72, 1), (4, 24, 16, 128, 32, 32, 16, 1), (8, 20, 0, 72, 16, 16, 16, 1),\n"
        "                                       (8, 16, 0, 64, 32, 64, 64, 2), (8, 24, 0, 128, 32, 64, 32, 2), (3, 10, 0, 40, 11, 30, 68, 2)]:\n"
        "    Cin = C0 + C1\n"
        "    Do, Ho, Wo = (D - 1) // st + 1, (H - 1) // st + 1, (W - 1) // st + 1\n"
        "    x0 = torch.randn(N, C0, D, H, W, generator=g).cuda(); s0 = (torch.rand(N, C0, 2, generator=g) + 0.5).cuda()\n"
        "    x1 = torch.randn(N, C1, D, H, W, generator=g).cuda() if C1 else None; s1 = (torch.rand(N, C1, 2, generator=g) + 0.5).cuda() if C1 else None\n"
        "    w = (torch.randn(Cout, Cin, 27, generator=g) * 0.1).numpy(); b = torch.randn(Cout, generator=g).cuda()\n"
        "    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, st)\n"
        "    wf = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout), np.float32)\n"
        "    lib.wmh_conv3d_k3_pack_weights(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))\n"
        "    wfd = torch.from_numpy(wf).cuda(); y = torch.zeros(N, Cout, Do, Ho, Wo, device='cuda')\n"
        "    npart = lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo); part = torch.zeros(N, Cout, npart, 4, device='cuda')\n"
        "    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, _lib.ptr(x0), _lib.ptr(s0), C0, _lib.ptr(x1), _lib.ptr(s1), C1, _lib.ptr(wfd), _lib.ptr(b), _lib.ptr(y),\n"
        "               _lib.ptr(part), None, N, D, H, W, Cout, st, 0.01, None), 'conv')\n"
        "    torch.cuda.synchronize(); res_y.append(y.cpu().numpy().ravel()); res_p.append(part.cpu().numpy().ravel()); res_p.append(np.array([cfg], np.float32))\n"
        "np.save(sys.argv[1], np.concatenate(res_y)); np.save(sys.argv[2], np.concatenate(res_p))\n") % root
    outs = []
    from deepwmh_amd import _lib
    for ws in ('0', '1', 'product'):
        f1, f2 = str(tmp_path / ('y%s.npy' % ws)), str(tmp_path / ('p%s.npy' % ws))
        env = dict(os.environ) if ws == 'product' else dict(os.environ, WMH_LIB_PATH=_lib.DIAG_LIB_PATH, WMH_WS=ws, WMH_KSPLIT='1')
        r = subprocess.run([sys.executable, '-c', script, f1, f2], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(f1), np.load(f2)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert float(np.abs(outs[0][0]).max()) > 0.1
    # the product library is a separate compilation of the same sources (no switches: wave-specialised; no slab memory is passed here, so
    # it cannot split K -- conv3d_k3.hip: "no slab memory given: run unsplit"): same bits, outputs and moment records
    assert outs[2][0].shape == outs[1][0].shape
    assert np.array_equal(outs[2][0], outs[1][0]) and np.array_equal(outs[2][1], outs[1][1])


BF16_SMALL_CASES = [(2, 32, 0, 64, 8, 8, 8, 1), (1, 48, 16, 32, 5, 9, 7, 1), (2, 32, 0, 64, 16, 16, 16, 2), (1, 16, 0, 32, 9, 13, 15, 2),
                    (2, 32, 0, 128, 8, 32, 32, 2), (1, 16, 0, 64, 7, 20, 27, 2)]


@pytest.mark.parametrize('case', BF16_SMALL_CASES, ids=[str(c) for c in BF16_SMALL_CASES])
def test_conv3d_k3_bf16_small_rows_and_strided(lib, case):
    """bf16 operands on the deep levels: rows of 8 voxels (configuration 4) and the stride-2 convolutions whose output rows
    are 8 or 16 voxels (configurations 5 and 3; D, H, W = the INPUT extent), against the convolution of bf16-rounded operands."""
    from deepwmh_amd import _lib
    N, C0, C1, Cout, D, H, W, s = case
    g = torch.Generator().manual_seed(98)
    Cin = C0 + C1
    x = torch.randn(N, Cin, D, H, W, generator=g) * 1.5 + 0.3
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    r = lambda t: t.to(torch.bfloat16).float()
    ref = F.conv3d(r(_lrelu_affine(x, ss)), r(w), b, padding=1, stride=s)
    Do, Ho, Wo = ref.shape[2:]
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, s)
    assert cfg in (3, 4, 5) and lib.wmh_conv3d_k3_bf16_supported(cfg)
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cin, Cout), np.uint16)
    wn = w.numpy()
    assert lib.wmh_conv3d_k3_pack_weights_bf16(cfg, wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, Do, Ho, Wo)
    y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    mr = torch.full((N, Cout, 2), float('nan'), device='cuda'); ss_out = torch.empty(N, Cout, 2, device='cuda')
    x0 = x[:, :C0].contiguous().cuda()
    x1 = x[:, C0:].contiguous().cuda() if C1 else None
    s0 = ss[:, :C0].contiguous().cuda()
    s1 = ss[:, C0:].contiguous().cuda() if C1 else None
    wfd, bd = dev(wf.view(np.int16)), b.cuda()
    _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(x0), P(s0), C0, P(x1), P(s1), C1, P(wfd), P(bd), P(y), P(part),
                                          N, D, H, W, Cout, SLOPE, None), 'conv bf16')
    _lib.check(lib.wmh_instnorm_finalize(P(part), npart, None, None, 1e-5, P(ss_out), P(mr), N, Cout, None), 'fin')
    torch.cuda.synchronize()
    yh = y.cpu()
    assert not torch.isnan(yh).any()
    scale = float(ref.abs().max())
    assert float((yh - ref).abs().max()) <= 5e-4 * scale + 1e-5
    mean = yh.mean((2, 3, 4)).double()
    assert float((mr.cpu().double()[..., 0] - mean).abs().max()) <= 1e-5 * (1 + float(mean.abs().max()))
    # the split-K form of these configurations (wmh_conv3d_k3_fwd_bf16_splitk): the channel reduction in slabs, summed in fixed order with the
    # bias, moments as one record per (n, c) -- same result up to the summation order, same statistics, deterministic
    nws = lib.wmh_conv3d_k3_bf16_workspace_floats(cfg, Cin, N, D, H, W, Cout)
    ws = torch.empty(max(nws, 1), device='cuda')
    outs = []
    for _ in range(2):
        y2 = torch.full_like(y, float('nan')); part2 = torch.full_like(part, float('nan'))
        mr2 = torch.empty_like(mr); ss2 = torch.empty_like(ss_out)
        _lib.check(lib.wmh_conv3d_k3_fwd_bf16_splitk(cfg, P(x0), P(s0), C0, P(x1), P(s1), C1, P(wfd), P(bd), P(y2), P(part2), P(ws) if nws else None,
                                                     N, D, H, W, Cout, SLOPE, None), 'conv bf16 split-K')
        _lib.check(lib.wmh_instnorm_finalize(P(part2), npart, None, None, 1e-5, P(ss2), P(mr2), N, Cout, None), 'fin')
        torch.cuda.synchronize()
        outs.append((y2.cpu(), ss2.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0] - yh).abs().max()) <= 2e-5 * scale + 1e-6
    assert float((outs[0][1] - ss_out.cpu()).abs().max()) <= 1e-4 * (1 + float(ss_out.abs().max()))
    if nws:
        assert Cin >= 32          # the cases that split are the ones with enough channel chunks


def test_bf16_wave_specialised_form_is_bit_identical(tmp_path):
    """conv3d_k3_bf16_ws.hip (loader / consumer waves, persistent: what the product library runs) against the one-role bf16 kernel
    (libdeepwmh_amd_diag.so with WMH_BF16_WS=0 -- variant switches exist in the diagnostic library only): same arithmetic and k order,
    so outputs and InstanceNorm partials must agree bit for bit.  Two sources, channel tails, ragged tiles, both tile geometries
    (32-wide rows and 16-wide rows)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from deepwmh_amd import _lib\n"
        "lib = _lib.load(); P = _lib.ptr\n"
        "g = torch.Generator().manual_seed(3)\n"
        "outs = []\n"
        "for (N, C0, C1, Cout, D, H, W) in [(2, 12, 0, 40, 9, 20, 70), (3, 32, 24, 32, 8, 16, 33), (2, 48, 0, 72, 6, 18, 16), (5, 16, 16, 64, 2, 33, 12)]:\n"
        "    Cin = C0 + C1\n"
        "    x0 = torch.randn(N, C0, D, H, W, generator=g).cuda(); s0 = (torch.rand(N, C0, 2, generator=g) + 0.5).cuda()\n"
        "    x1 = torch.randn(N, C1, D, H, W, generator=g).cuda() if C1 else None; s1 = (torch.rand(N, C1, 2, generator=g) + 0.5).cuda() if C1 else None\n"
        "    w = (torch.randn(Cout, Cin, 27, generator=g) * 0.1).cuda(); b = torch.randn(Cout, generator=g).cuda()\n"
        "    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1)\n"
        "    assert lib.wmh_conv3d_k3_bf16_supported(cfg), cfg\n"
        "    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cin, Cout), dtype=torch.int16, device='cuda')\n"
        "    _lib.check(lib.wmh_conv3d_k3_pack_weights_bf16_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None))\n"
        "    y = torch.zeros(N, Cout, D, H, W, device='cuda')\n"
        "    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, D, H, W); part = torch.zeros(N, Cout, npart, 4, device='cuda')\n"
        "    _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(x0), P(s0), C0, P(x1), P(s1), C1, P(wf), P(b), P(y), P(part), N, D, H, W, Cout, 0.01, None), 'conv')\n"
        "    torch.cuda.synchronize(); outs += [y.cpu().numpy().ravel(), part.cpu().numpy().ravel(), np.array([cfg], np.float32)]\n"
        "np.save(sys.argv[1], np.concatenate(outs))\n") % root
    res = []
    from deepwmh_amd import _lib
    for ws in ('0', '1'):
        f1 = str(tmp_path / ('y%s.npy' % ws))
        env = dict(os.environ, WMH_LIB_PATH=_lib.DIAG_LIB_PATH, WMH_BF16_WS='0') if ws == '0' else dict(os.environ)     # '1': the product library
        r = subprocess.run([sys.executable, '-c', script, f1], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(f1))
    assert res[0].shape == res[1].shape and np.array_equal(res[0], res[1])
    assert float(np.abs(res[0]).max()) > 0.1
