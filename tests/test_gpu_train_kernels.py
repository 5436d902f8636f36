"""GPU parity tests of the training-step kernels against torch autograd (CPU fp32) of the same ops."""
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
    return _lib.load()


def P(t):
    from deepwmh_amd import _lib
    return _lib.ptr(t)


def check(rc, what=''):
    from deepwmh_amd import _lib
    return _lib.check(rc, what)


def _act(x, ss):
    if ss is None:
        return x
    return F.leaky_relu(x * ss[..., 0][:, :, None, None, None] + ss[..., 1][:, :, None, None, None], SLOPE)


def _run_conv(lib, cfg, x, ss, wf, Cout, stride, N, D, H, W):
    Cin = x.shape[1]
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    nws = lib.wmh_conv3d_k3_workspace_floats(cfg, Cin, N, D, H, W, Cout, stride)
    ws = torch.empty(nws, device='cuda') if nws else None
    check(lib.wmh_conv3d_k3_fwd(cfg, P(x), P(ss), Cin, None, None, 0, P(wf), None, P(y), None, P(ws), N, D, H, W, Cout, stride,
                                SLOPE, None), 'conv fwd')
    return y


@pytest.mark.parametrize('shape', [(2, 8, 16, 8, 8, 32, 1), (1, 6, 40, 5, 9, 33, 1), (2, 16, 8, 8, 8, 8, 1), (1, 32, 32, 4, 4, 4, 1),
                                   (1, 8, 16, 16, 16, 32, 2), (2, 16, 32, 8, 8, 16, 2), (1, 12, 8, 9, 7, 13, 2)])
def test_device_pack_and_data_gradient(lib, shape):
    # forward image packed on the device == host pack; data gradient = forward kernel on dy with the mode-1 image
    N, Cin, Cout, D, H, W, s = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, D, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1
    y = F.conv3d(x, w, None, stride=s, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, s)
    wd = w.cuda()
    nf = lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout)
    wf_dev = torch.full((nf,), float('nan'), device='cuda')
    check(lib.wmh_conv3d_k3_pack_weights_device(cfg, P(wd), Cin, Cout, 0, Cin, 0, P(wf_dev), None), 'pack0')
    wf_host = np.empty(nf, np.float32)
    wn = w.numpy()
    lib.wmh_conv3d_k3_pack_weights(cfg, wn.ctypes.data_as(C.c_void_p), Cin, Cout, wf_host.ctypes.data_as(C.c_void_p))
    torch.cuda.synchronize()
    assert np.array_equal(wf_dev.cpu().numpy(), wf_host)
    # data gradient
    dyd = dy.cuda()
    Do, Ho, Wo = y.shape[2:]
    if s == 2:
        up = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
        check(lib.wmh_upsample_zero2(P(dyd), P(up), N, Cout, Do, Ho, Wo, D, H, W, None), 'up')
        src = up
    else:
        src = dyd
    cfgb = lib.wmh_conv3d_k3_config(Cout, Cin, D, H, W, 1)
    wfb = torch.full((lib.wmh_conv3d_k3_packed_floats(cfgb, Cout, Cin),), float('nan'), device='cuda')
    check(lib.wmh_conv3d_k3_pack_weights_device(cfgb, P(wd), Cin, Cout, 0, Cin, 1, P(wfb), None), 'pack1')
    dx = _run_conv(lib, cfgb, src, None, wfb, Cin, 1, N, D, H, W)
    torch.cuda.synchronize()
    err = float((dx.cpu() - x.grad).abs().max())
    assert err <= 1e-5 * (1 + float(x.grad.abs().max())), err


WG_CASES = [(2, 8, 0, 16, 8, 8, 32, 1, True, False), (1, 6, 0, 40, 5, 9, 33, 1, False, False), (2, 4, 12, 8, 4, 8, 32, 1, True, True),
            (2, 16, 0, 8, 8, 8, 16, 1, True, False), (3, 32, 0, 32, 8, 8, 8, 1, True, False), (5, 40, 0, 36, 4, 4, 4, 1, True, False),
            (1, 8, 0, 16, 16, 16, 64, 2, True, False), (2, 16, 0, 32, 8, 8, 32, 2, False, False), (1, 12, 0, 8, 9, 7, 13, 2, True, False),
            (1, 1, 0, 8, 8, 8, 32, 1, False, False), (1, 64, 0, 64, 4, 8, 32, 1, True, False)]


@pytest.mark.parametrize('case', WG_CASES, ids=[str(c) for c in WG_CASES])
def test_conv3d_wgrad(lib, case):
    N, C0, C1, Cout, D, H, W, s, use0, use1 = case
    g = torch.Generator().manual_seed(31)
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W, generator=g)
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    ss0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1) if use0 else None
    ss1 = torch.stack([0.5 + torch.rand(N, C1, generator=g), 0.3 * torch.randn(N, C1, generator=g)], -1) if use1 else None
    xin = _act(x0, ss0)
    if C1:
        xin = torch.cat((xin, _act(x1, ss1)), 1)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1).requires_grad_()
    y = F.conv3d(xin, w, None, stride=s, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    d = [t.cuda() if t is not None else None for t in (x0, ss0, x1, ss1)]
    dyd = dy.cuda()
    dW = torch.full((Cout, Cin, 27), float('nan'), device='cuda')
    ws = torch.empty(lib.wmh_conv3d_k3_wgrad_workspace_floats(Cin, Cout, N, D, H, W, s), device='cuda')
    check(lib.wmh_conv3d_k3_wgrad(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(dyd), P(dW), P(ws), N, D, H, W, Cout, s, SLOPE, None), 'wgrad')
    torch.cuda.synchronize()
    got = dW.cpu().view(Cout, Cin, 3, 3, 3)
    assert not torch.isnan(got).any()
    err = float((got - w.grad).abs().max())
    assert err <= 2e-5 * (1 + float(w.grad.abs().max())), err


@pytest.mark.parametrize('K,shape', [(2, (2, 9, 10, 11)), (3, (1, 16, 16, 16)), (2, (3, 4, 4, 4))])
def test_dice_ce_loss(lib, K, shape):
    from oracle.training import dc_and_ce_loss
    g = torch.Generator().manual_seed(41)
    N = shape[0]
    sp = shape[1:]
    logits = (torch.randn(N, K, *sp, generator=g) * 2).requires_grad_()
    target = torch.randint(0, K, (N, 1) + tuple(sp), generator=g)
    wgt = 0.2667
    loss = wgt * dc_and_ce_loss(logits, target.float())
    loss.backward()
    S = int(np.prod(sp))
    ld, td = logits.detach().cuda(), target.to(torch.uint8).cuda()
    dl = torch.full((N, K, S), float('nan'), device='cuda')
    acc = torch.zeros(1, dtype=torch.float64, device='cuda')
    ws = torch.empty(lib.wmh_dice_ce_loss_workspace_doubles(), dtype=torch.float64, device='cuda')
    check(lib.wmh_dice_ce_loss_fwd_bwd(P(ld), P(td), N, K, S, wgt, 1e-5, P(dl), P(acc), P(ws), None), 'loss')
    torch.cuda.synchronize()
    assert abs(float(acc.cpu()) - float(loss)) <= 1e-5 * (1 + abs(float(loss)))
    err = float((dl.cpu().view_as(logits) - logits.grad).abs().max())
    assert err <= 1e-6 + 1e-4 * float(logits.grad.abs().max()), err


@pytest.mark.parametrize('K,shape', [(2, (2, 9, 10, 11)), (3, (1, 16, 16, 16)), (4, (3, 4, 4, 4)), (2, (2, 64, 64, 64))])
def test_online_eval_counts(lib, K, shape):
    """The hard tp / fp / fn counts behind nnU-Net's model_best criterion against the oracle's run_online_evaluation: exact integers, added
    to what the counters hold, ties to the first class (torch.argmax)."""
    from oracle.training import run_online_evaluation
    g = torch.Generator().manual_seed(43)
    N, sp = shape[0], shape[1:]
    S = int(np.prod(sp))
    counts = torch.zeros(3 * (K - 1), dtype=torch.int64, device='cuda')
    want = np.zeros(3 * (K - 1))
    for b in range(2):
        logits = torch.randn(N, K, *sp, generator=g)
        logits[0, :, 0, 0, :3] = 0.25                                   # ties: class 0 wins
        target = torch.randint(0, K, (N, 1) + tuple(sp), generator=g)
        tp, fp, fn = run_online_evaluation(logits, target.float())
        want += np.concatenate([tp, fp, fn])
        ld, td = logits.cuda(), target.to(torch.uint8).cuda()           # (kept alive: the launch is asynchronous)
        check(lib.wmh_online_eval_counts(P(ld), P(td), N, K, S, P(counts), None), 'online_eval_counts')
        torch.cuda.synchronize()
    assert np.array_equal(counts.cpu().numpy(), want.astype(np.int64))


@pytest.mark.parametrize('with_g1,with_g2,with_head', [(True, False, False), (True, True, False), (False, False, True), (True, False, True)])
def test_in_lrelu_bwd(lib, with_g1, with_g2, with_head):
    g = torch.Generator().manual_seed(51)
    N, Cc, K, sp = 2, 6, 2, (6, 7, 8)
    S = int(np.prod(sp))
    y = (torch.randn(N, Cc, *sp, generator=g) * 2 + 0.5).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(Cc, generator=g)).requires_grad_()
    wseg = torch.randn(K, Cc, generator=g).requires_grad_()
    a = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), SLOPE)
    g1 = torch.randn(a.shape, generator=g) if with_g1 else None
    g2 = torch.randn(a.shape, generator=g) if with_g2 else None
    dl = torch.randn(N, K, *sp, generator=g) if with_head else None
    obj = 0
    if with_g1: obj = obj + (a * g1).sum()
    if with_g2: obj = obj + (a * g2).sum()
    if with_head: obj = obj + (torch.einsum('kc,ncxyz->nkxyz', wseg, a) * dl).sum()
    obj.backward()
    yd = y.detach().cuda()
    ss = torch.empty(N, Cc, 2, device='cuda'); mr = torch.empty(N, Cc, 2, device='cuda')
    gd, bd = gamma.detach().cuda(), beta.detach().cuda()
    check(lib.wmh_instnorm_stats(P(yd), P(gd), P(bd), 1e-5, P(ss), P(mr), N, Cc, S, None), 'stats')
    dy = torch.full((N, Cc, S), float('nan'), device='cuda')
    dgam = torch.full((Cc,), float('nan'), device='cuda'); dbet = torch.full((Cc,), float('nan'), device='cuda')
    dws = torch.full((K, Cc), float('nan'), device='cuda')
    sums = torch.empty(lib.wmh_in_lrelu_bwd_scratch_doubles(N, Cc, S), dtype=torch.float64, device='cuda')
    dbias = torch.full((Cc,), float('nan'), device='cuda')
    g1d = g1.cuda() if with_g1 else None; g2d = g2.cuda() if with_g2 else None
    dld = dl.cuda() if with_head else None; wd = wseg.detach().cuda()
    check(lib.wmh_in_lrelu_bwd(P(yd), P(ss), P(mr), P(gd), P(g1d), P(g2d), P(dld), P(wd), K, P(dy), P(dgam), P(dbet), P(dws), P(dbias), P(sums),
                               N, Cc, S, SLOPE, None), 'in bwd')
    torch.cuda.synchronize()
    sc = float(y.grad.abs().max())
    assert float((dy.cpu().view_as(y) - y.grad).abs().max()) <= 2e-5 * (1 + sc)
    assert float((dbias.cpu() - y.grad.sum((0, 2, 3, 4))).abs().max()) <= 1e-3 * (1 + sc)
    assert float((dgam.cpu() - gamma.grad).abs().max()) <= 1e-4 * (1 + float(gamma.grad.abs().max()))
    assert float((dbet.cpu() - beta.grad).abs().max()) <= 1e-4 * (1 + float(beta.grad.abs().max()))
    if with_head:
        assert float((dws.cpu() - wseg.grad).abs().max()) <= 1e-4 * (1 + float(wseg.grad.abs().max()))


@pytest.mark.parametrize('shape', [(2, 16, 8, 4, 4, 4, True), (1, 8, 12, 3, 6, 5, False)])
def test_convtranspose3d_bwd(lib, shape):
    N, Cin, Cout, D, H, W, use_ss = shape
    g = torch.Generator().manual_seed(61)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1) if use_ss else None
    a = _act(x, ss).requires_grad_()
    w = (torch.randn(Cin, Cout, 2, 2, 2, generator=g) * 0.2).requires_grad_()
    out = F.conv_transpose3d(a, w, None, stride=2)
    dY = torch.randn(out.shape, generator=g)
    out.backward(dY)
    xd, ssd, wd, dYd = x.cuda(), (ss.cuda() if use_ss else None), w.detach().cuda(), dY.cuda()
    dX = torch.full((N, Cin, D, H, W), float('nan'), device='cuda'); dW = torch.full((Cin, Cout, 8), float('nan'), device='cuda')
    check(lib.wmh_convtranspose3d_k2s2_bwd(P(xd), P(ssd), P(wd), P(dYd), P(dX), P(dW), N, Cin, Cout, D, H, W, SLOPE, None), 'convT bwd')
    torch.cuda.synchronize()
    assert float((dX.cpu() - a.grad).abs().max()) <= 1e-5 * (1 + float(a.grad.abs().max()))
    assert float((dW.cpu().view_as(w) - w.grad).abs().max()) <= 2e-5 * (1 + float(w.grad.abs().max()))


@pytest.mark.parametrize('shape', [(2, 16, 8, 4, 4, 4, True), (1, 8, 12, 3, 6, 5, False), (2, 40, 72, 4, 8, 16, True),
                                   (1, 64, 32, 6, 10, 20, True)])
def test_convtranspose3d_bwd_mfma(lib, shape):
    """MFMA forms: dW with voxels as the K dimension; dX as a stride-2 convolution (pack mode 2) of dY."""
    N, Cin, Cout, D, H, W, use_ss = shape
    g = torch.Generator().manual_seed(62)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    ss = torch.stack([0.5 + torch.rand(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)], -1) if use_ss else None
    a = _act(x, ss).requires_grad_()
    w = (torch.randn(Cin, Cout, 2, 2, 2, generator=g) * 0.2).requires_grad_()
    out = F.conv_transpose3d(a, w, None, stride=2)
    dY = torch.randn(out.shape, generator=g)
    out.backward(dY)
    xd, ssd, wd, dYd = x.cuda(), (ss.cuda() if use_ss else None), w.detach().cuda(), dY.cuda()
    dW = torch.full((Cin, Cout, 8), float('nan'), device='cuda')
    ws = torch.empty(max(1, lib.wmh_convtranspose3d_k2s2_wgrad_workspace_floats(Cin, Cout, N, D, H, W)), device='cuda')
    check(lib.wmh_convtranspose3d_k2s2_wgrad(P(xd), P(ssd), P(dYd), P(dW), P(ws), N, Cin, Cout, D, H, W, SLOPE, None), 'convT wgrad')
    torch.cuda.synchronize()
    assert float((dW.cpu().view_as(w) - w.grad).abs().max()) <= 2e-5 * (1 + float(w.grad.abs().max()))
    dW2 = torch.full((Cin, Cout, 8), float('nan'), device='cuda')
    check(lib.wmh_convtranspose3d_k2s2_wgrad(P(xd), P(ssd), P(dYd), P(dW2), P(ws), N, Cin, Cout, D, H, W, SLOPE, None), 'convT wgrad')
    assert torch.equal(dW, dW2)                                              # fixed-order reduction
    cfg = check(lib.wmh_conv3d_k3_config(Cout, Cin, 2 * D, 2 * H, 2 * W, 2), 'config')
    wf = torch.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cout, Cin), device='cuda')
    check(lib.wmh_conv3d_k3_pack_weights_device(cfg, P(wd), Cout, Cin, 0, Cout, 2, P(wf), None), 'pack mode 2')
    dX = torch.full((N, Cin, D, H, W), float('nan'), device='cuda')
    nws = lib.wmh_conv3d_k3_workspace_floats(cfg, Cout, N, 2 * D, 2 * H, 2 * W, Cin, 2)
    cws = torch.empty(max(1, nws), device='cuda')
    check(lib.wmh_conv3d_k3_fwd(cfg, P(dYd), None, Cout, None, None, 0, P(wf), None, P(dX), None, P(cws), N, 2 * D, 2 * H, 2 * W, Cin, 2,
                                SLOPE, None), 'conv fwd (convT dX)')
    torch.cuda.synchronize()
    assert float((dX.cpu() - a.grad).abs().max()) <= 1e-5 * (1 + float(a.grad.abs().max()))


def test_convtranspose3d_device_pack_matches_host(lib):
    g = torch.Generator().manual_seed(63)
    for Cin, Cout, kt in [(16, 8, 2), (40, 24, 2), (12, 20, 1)]:
        w = torch.randn(Cin, Cout, kt, 2, 2, generator=g)
        n = lib.wmh_convtranspose3d_packed_floats(Cin, Cout, kt)
        host = np.empty(n, dtype=np.float32)
        wn = w.numpy()
        check(lib.wmh_convtranspose3d_pack_weights(wn.ctypes.data_as(C.c_void_p), Cin, Cout, kt, host.ctypes.data_as(C.c_void_p)), 'pack')
        wd = w.cuda(); dev = torch.empty(n, device='cuda')
        check(lib.wmh_convtranspose3d_pack_weights_device(P(wd), Cin, Cout, kt, P(dev), None), 'pack device')
        assert np.array_equal(dev.cpu().numpy(), host)


@pytest.mark.parametrize('shape', [(2, 8, 16, 8, 8, 8), (1, 40, 20, 9, 11, 13), (1, 32, 64, 12, 10, 70), (2, 3, 5, 4, 6, 5)])
def test_conv3d_stride2_dgrad_gather(lib, shape):
    """dx of Conv3d(k3, s2, p1) in gather form against autograd (even and odd extents, channel tails, several x tiles)."""
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(81)
    x = torch.randn(N, Cin, D, H, W, generator=g).requires_grad_()
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1
    y = F.conv3d(x, w, None, stride=2, padding=1)
    dY = torch.randn(y.shape, generator=g)
    y.backward(dY)
    wd, dYd = w.cuda(), dY.cuda()
    wp = torch.empty(lib.wmh_conv3d_k3s2_dgrad_packed_floats(Cin, Cout), device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad_pack_weights_device(P(wd), Cin, Cout, P(wp), None), 'dgrad pack')
    dX = torch.full((N, Cin, D, H, W), float('nan'), device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad(P(dYd), P(wp), P(dX), N, Cin, Cout, D, H, W, None), 'k3s2 dgrad')
    torch.cuda.synchronize()
    assert float((dX.cpu() - x.grad).abs().max()) <= 1e-5 * (1 + float(x.grad.abs().max()))


@pytest.mark.parametrize('shape', [(2, 8, 16, 8, 8, 8), (1, 40, 20, 9, 11, 13), (1, 32, 64, 12, 10, 70), (2, 3, 5, 4, 6, 5), (2, 32, 64, 16, 16, 64)])
def test_conv3d_stride2_dgrad_f16x2_pairs(lib, shape):
    """dx of Conv3d(k3, s2, p1) from fp16 pairs against float64 autograd (even and odd extents, channel tails, several x tiles); the
    exact-fp32 gather kernel's own distance is printed beside it."""
    N, Cin, Cout, D, H, W = shape
    g = torch.Generator().manual_seed(82)
    x = torch.zeros(N, Cin, D, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * torch.exp(torch.empty(Cout, 1, 1, 1, 1).uniform_(-5, 0, generator=g))
    y = F.conv3d(x, w.double(), None, stride=2, padding=1)
    dY = torch.randn(y.shape, generator=g) * torch.exp(torch.empty(1, Cout, 1, 1, 1).uniform_(-3, 3, generator=g))
    y.backward(dY.double())
    ref = x.grad
    wd, dYd = w.cuda(), dY.cuda()
    wp = torch.zeros(lib.wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(Cin, Cout), dtype=torch.int32, device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device(P(wd), Cin, Cout, P(wp), None), 'dgrad pack f16x2')
    dX = torch.full((N, Cin, D, H, W), float('nan'), device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad_f16x2(P(dYd), P(wp), P(dX), N, Cin, Cout, D, H, W, None), 'k3s2 dgrad f16x2')
    wp32 = torch.empty(lib.wmh_conv3d_k3s2_dgrad_packed_floats(Cin, Cout), device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad_pack_weights_device(P(wd), Cin, Cout, P(wp32), None), 'dgrad pack')
    dX32 = torch.full((N, Cin, D, H, W), float('nan'), device='cuda')
    check(lib.wmh_conv3d_k3s2_dgrad(P(dYd), P(wp32), P(dX32), N, Cin, Cout, D, H, W, None), 'k3s2 dgrad')
    torch.cuda.synchronize()
    got = dX.cpu().double()
    assert not torch.isnan(got).any()
    sc = float(ref.abs().max())
    e16, r16 = float((got - ref).abs().max()) / sc, float((got - ref).norm() / ref.norm())
    e32, r32 = float((dX32.cpu().double() - ref).abs().max()) / sc, float((dX32.cpu().double() - ref).norm() / ref.norm())
    print('f16x2 max err / max|ref| %.2e, rel. L2 %.2e  (exact-fp32 kernel %.2e / %.2e)' % (e16, r16, e32, r32))
    assert e16 <= 2e-6 and r16 <= 1e-6


F16X2_WG = [(2, 16, 0, 8, 8, 8, 32), (1, 32, 24, 40, 5, 9, 45), (2, 20, 0, 36, 6, 16, 16), (1, 8, 8, 72, 3, 17, 20), (2, 1, 0, 32, 4, 8, 32),
            (2, 64, 0, 128, 8, 32, 32), (1, 128, 0, 256, 6, 32, 32), (2, 32, 0, 32, 16, 32, 64), (3, 16, 0, 32, 4, 24, 40),
            (2, 48, 0, 64, 8, 8, 8), (1, 24, 16, 40, 5, 11, 10), (2, 320, 0, 64, 8, 8, 8)]      # rows of 8 voxels (round 5: the 8^3 level)


@pytest.mark.parametrize('shape', F16X2_WG, ids=[str(c) for c in F16X2_WG])
def test_conv3d_wgrad_f16x2_pairs(lib, shape):
    """Pair-form weight gradient (fp16 hi + lo per operand, conv3d_k3_wgrad_f16x2.hip) against float64 autograd: fp32-grade (the exact-fp32
    kernel's own distance to float64 is printed beside it); two sources, channel tails, ragged tiles, both tile geometries, the three
    tile orders (plain / part of a plane per step / several planes per step), bit-reproducible."""
    N, C0, C1, Cout, D, H, W = shape
    Cin = C0 + C1
    g = torch.Generator().manual_seed(92)
    x0 = torch.randn(N, C0, D, H, W, generator=g); s0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1)
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    s1 = torch.stack([0.5 + torch.rand(N, C1, generator=g), 0.3 * torch.randn(N, C1, generator=g)], -1) if C1 else None
    dy = torch.randn(N, Cout, D, H, W, generator=g) * torch.exp(torch.empty(N, Cout, 1, 1, 1).uniform_(-4, 3, generator=g))
    act = torch.cat([_act(x0, s0)] + ([_act(x1, s1)] if C1 else []), 1)            # fp32, as the kernel forms it
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(act.double(), w, None, padding=1).backward(dy.double())
    ref = w.grad.reshape(Cout, Cin, 27)
    assert lib.wmh_conv3d_k3_wgrad_f16x2_supported(Cin, Cout, N, D, H, W, 1) == 1
    assert lib.wmh_conv3d_k3_wgrad_f16x2_supported(Cin, Cout, N, D, H, 4, 1) == 0
    d = [t.cuda() if t is not None else None for t in (x0, s0, x1, s1, dy)]
    ws = torch.full((lib.wmh_conv3d_k3_wgrad_f16x2_workspace_floats(Cin, Cout, N, D, H, W),), float('nan'), device='cuda')
    outs = []
    for _ in range(2):
        dW = torch.full((Cout, Cin, 27), float('nan'), device='cuda')
        check(lib.wmh_conv3d_k3_wgrad_f16x2(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, SLOPE, None), 'wgrad f16x2')
        torch.cuda.synchronize()
        outs.append(dW.cpu())
    assert torch.equal(outs[0], outs[1])                                          # fixed-order reduction
    assert not torch.isnan(outs[0]).any()
    dW32 = torch.full((Cout, Cin, 27), float('nan'), device='cuda')
    ws32 = torch.empty(lib.wmh_conv3d_k3_wgrad_workspace_floats(Cin, Cout, N, D, H, W, 1), device='cuda')
    check(lib.wmh_conv3d_k3_wgrad(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW32), P(ws32), N, D, H, W, Cout, 1, SLOPE, None), 'wgrad')
    torch.cuda.synchronize()
    # per output channel: dy's scale varies over 7 e-folds between channels
    scale = ref.abs().amax((1, 2), keepdim=True)
    e16 = float(((outs[0].double() - ref).abs() / scale).max()); e32 = float(((dW32.cpu().double() - ref).abs() / scale).max())
    r16 = float((outs[0].double() - ref).norm() / ref.norm())
    print('f16x2 max err / channel max %.2e (exact-fp32 kernel %.2e), rel. L2 %.2e' % (e16, e32, r16))
    assert e16 <= 4e-6 and r16 <= 1e-6, (e16, e32, r16)


F16X2_WG_S2 = [(2, 16, 0, 32, 8, 8, 32), (1, 24, 0, 40, 7, 13, 30), (2, 32, 0, 64, 16, 32, 64), (1, 8, 8, 48, 9, 10, 52), (2, 64, 0, 128, 8, 16, 32),
               (3, 20, 0, 36, 4, 6, 25)]


@pytest.mark.parametrize('shape', F16X2_WG_S2, ids=[str(c) for c in F16X2_WG_S2])
def test_conv3d_wgrad_f16x2_pairs_stride2(lib, shape):
    """The pair-form weight gradient of the stride-2 convolution (round 5: halo rows parity-split in LDS, tiles of 16 x 2 x 2 outputs) against
    float64 autograd, beside the exact-fp32 kernel that served these layers before: even and odd input extents, ragged tiles, channel tails,
    two sources, the tile orders; bit-reproducible."""
    N, C0, C1, Cout, D, H, W = shape                  # the INPUT extent
    Cin = C0 + C1
    g = torch.Generator().manual_seed(93)
    x0 = torch.randn(N, C0, D, H, W, generator=g); s0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1)
    x1 = torch.randn(N, C1, D, H, W, generator=g) if C1 else None
    s1 = torch.stack([0.5 + torch.rand(N, C1, generator=g), 0.3 * torch.randn(N, C1, generator=g)], -1) if C1 else None
    Do, Ho, Wo = (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(N, Cout, Do, Ho, Wo, generator=g) * torch.exp(torch.empty(N, Cout, 1, 1, 1).uniform_(-4, 3, generator=g))
    act = torch.cat([_act(x0, s0)] + ([_act(x1, s1)] if C1 else []), 1)
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(act.double(), w, None, padding=1, stride=2).backward(dy.double())
    ref = w.grad.reshape(Cout, Cin, 27)
    assert lib.wmh_conv3d_k3_wgrad_f16x2_supported(Cin, Cout, N, D, H, W, 2) == 1
    assert lib.wmh_conv3d_k3_wgrad_f16x2_supported(Cin, Cout, N, D, H, 20, 2) == 0          # output rows of 10 voxels: the fp32 kernel
    d = [t.cuda() if t is not None else None for t in (x0, s0, x1, s1, dy)]
    ws = torch.full((lib.wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats(Cin, Cout, N, D, H, W),), float('nan'), device='cuda')
    outs = []
    for _ in range(2):
        dW = torch.full((Cout, Cin, 27), float('nan'), device='cuda')
        check(lib.wmh_conv3d_k3s2_wgrad_f16x2(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, SLOPE, None), 'wgrad f16x2 s2')
        torch.cuda.synchronize()
        outs.append(dW.cpu())
    assert torch.equal(outs[0], outs[1])                                          # fixed-order reduction
    assert not torch.isnan(outs[0]).any()
    dW32 = torch.full((Cout, Cin, 27), float('nan'), device='cuda')
    ws32 = torch.empty(lib.wmh_conv3d_k3_wgrad_workspace_floats(Cin, Cout, N, D, H, W, 2), device='cuda')
    check(lib.wmh_conv3d_k3_wgrad(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW32), P(ws32), N, D, H, W, Cout, 2, SLOPE, None), 'wgrad')
    torch.cuda.synchronize()
    scale = ref.abs().amax((1, 2), keepdim=True)
    e16 = float(((outs[0].double() - ref).abs() / scale).max()); e32 = float(((dW32.cpu().double() - ref).abs() / scale).max())
    r16 = float((outs[0].double() - ref).norm() / ref.norm())
    print('stride 2, f16x2 max err / channel max %.2e (exact-fp32 kernel %.2e), rel. L2 %.2e' % (e16, e32, r16))
    assert e16 <= 4e-6 and r16 <= 1e-6, (e16, e32, r16)


def test_clip_and_sgd_nesterov(lib):
    g = torch.Generator().manual_seed(71)
    n = 100003
    p = torch.randn(n, generator=g).requires_grad_()
    opt = torch.optim.SGD([p], 0.01, weight_decay=3e-5, momentum=0.99, nesterov=True)
    pd = p.detach().clone().cuda(); m = torch.zeros(n, device='cuda')
    clip = torch.empty(2, device='cuda'); ws = torch.empty(1024, dtype=torch.float64, device='cuda')
    for step in range(3):
        gr = torch.randn(n, generator=g) * (0.2 if step == 1 else 0.01)     # step 1 exceeds the clip norm of 12
        p.grad = gr.clone()
        norm = float(torch.nn.utils.clip_grad_norm_([p], 12))
        opt.step()
        gd = gr.cuda()
        check(lib.wmh_grad_clip_coef(P(gd), n, 12.0, P(clip), P(ws), None), 'clip')
        check(lib.wmh_sgd_nesterov_step(P(pd), P(gd), P(m), n, 0.01, 0.99, 3e-5, P(clip), 1 if step == 0 else 0, None), 'sgd')
        torch.cuda.synchronize()
        assert abs(float(clip[0].cpu()) - norm) <= 1e-4 * norm
        assert float((pd.cpu() - p.detach()).abs().max()) <= 1e-6
    assert float(clip[1].cpu()) == 1.0
