"""The decoder stage's transposed convolution composed with the `up` half of the 3x3x3 convolution behind it (csrc/conv3d_up_f16x2.hip)
against float64 torch: conv3d(cat(conv_transpose3d(act(x_low)), act(skip))) -- nnU-Net's `tu` + first decoder convolution
[Generic_UNet.forward; reached through /root/reference/deepwmh/main/predict.py:153-156]."""
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


def _act(x, ss):
    return F.leaky_relu(x.double() * ss[..., 0, None, None, None].double() + ss[..., 1, None, None, None].double(), SLOPE)


# (N, Clow, Cup, Cskip, Cout, Dl, Hl, Wl)
CASES = [(1, 8, 8, 8, 32, 4, 4, 16), (2, 20, 12, 12, 32, 4, 8, 32), (1, 64, 32, 32, 32, 8, 8, 32), (2, 16, 16, 16, 64, 8, 8, 16), (3, 9, 5, 8, 32, 4, 12, 48)]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_composed_up_convolution_matches_float64(lib, case):
    from deepwmh_amd import _lib
    P = _lib.ptr
    N, Cl, Cu, Cs, Cout, Dl, Hl, Wl = case
    D, H, W = 2 * Dl, 2 * Hl, 2 * Wl
    assert lib.wmh_conv3d_up_supported_f16x2(Cl, Cu, Cout, D, H, W)
    g = torch.Generator().manual_seed(7)
    xl = torch.randn(N, Cl, Dl, Hl, Wl, generator=g) * 1.3 + 0.2
    ssl = torch.stack([0.6 + torch.rand(N, Cl, generator=g), 0.3 * torch.randn(N, Cl, generator=g)], -1)
    skip = torch.randn(N, Cs, D, H, W, generator=g)
    sss = torch.stack([0.6 + torch.rand(N, Cs, generator=g), 0.3 * torch.randn(N, Cs, generator=g)], -1)
    wt = torch.randn(Cl, Cu, 2, 2, 2, generator=g) * (1.0 / Cl) ** 0.5
    w3 = torch.randn(Cout, Cu + Cs, 3, 3, 3, generator=g) * (2.0 / (27 * (Cu + Cs))) ** 0.5
    bias = 0.1 * torch.randn(Cout, generator=g)
    up = F.conv_transpose3d(_act(xl, ssl), wt.double(), stride=2)
    ref_up = F.conv3d(up, w3[:, :Cu].double(), None, padding=1)
    ref = F.conv3d(torch.cat([up, _act(skip, sss)], 1), w3.double(), bias.double(), padding=1)

    # the composed weight image (host, double precision)
    wimg = np.zeros(lib.wmh_conv3d_up_packed_bytes_f16x2(Cl, Cout), np.uint8)
    w3n, wtn = np.ascontiguousarray(w3.numpy().reshape(Cout, Cu + Cs, 27)), np.ascontiguousarray(wt.numpy().reshape(Cl, Cu, 8))
    _lib.check(lib.wmh_conv3d_up_pack_weights_f16x2(w3n.ctypes.data_as(C.c_void_p), Cu + Cs, Cu, wtn.ctypes.data_as(C.c_void_p), Cl, Cout,
                                                    wimg.ctypes.data_as(C.c_void_p)), 'pack up')
    wimg_d = torch.from_numpy(wimg).cuda()
    xl_d, ssl_d = xl.cuda(), ssl.contiguous().cuda()
    npart = lib.wmh_conv3d_up_num_partials_f16x2(D, H, W)

    # (a) the composed part alone (accumulate = 0) against conv(tu(x)) restricted to the up channels
    y = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_up_fwd_f16x2(P(xl_d), P(ssl_d), Cl, P(wimg_d), P(y), P(part), N, D, H, W, Cout, SLOPE, 0, None), 'up fwd')
    torch.cuda.synchronize()
    e = float((y.cpu().double() - ref_up).norm() / ref_up.norm())
    emax = float((y.cpu().double() - ref_up).abs().max() / ref_up.abs().max())
    print('composed part alone: rel L2 %.3g, max err / max|ref| %.3g' % (e, emax))
    assert e <= 2e-6 and emax <= 1e-5

    # (b) the whole stage: the pair kernel on the skip channels (y = bias + conv(skip)), then the composed part on top with the moments
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cs, Cout, D, H, W, 1), Cout, W)
    assert lib.wmh_conv3d_k3_f16x2_supported(cfg)
    w3_d = w3.reshape(Cout, Cu + Cs, 27).contiguous().cuda()
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cs, Cout), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w3_d), Cu + Cs, Cout, Cu, Cs, 0, P(wf), None), 'pack skip slice')
    skip_d, sss_d, bias_d = skip.cuda(), sss.contiguous().cuda(), bias.cuda()
    nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, Cs, N, D, H, W, Cout)
    ws = torch.empty(max(1, nws), device='cuda')
    y.fill_(float('nan'))
    _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(skip_d), P(sss_d), Cs, None, None, 0, P(wf), P(bias_d), P(y), None, P(ws) if nws else None,
                                           N, D, H, W, Cout, SLOPE, None), 'skip conv')
    _lib.check(lib.wmh_conv3d_up_fwd_f16x2(P(xl_d), P(ssl_d), Cl, P(wimg_d), P(y), P(part), N, D, H, W, Cout, SLOPE, 1, None), 'up fwd accumulate')
    ss_out = torch.empty(N, Cout, 2, device='cuda'); mr = torch.empty(N, Cout, 2, device='cuda')
    _lib.check(lib.wmh_instnorm_finalize(P(part), npart, None, None, 1e-5, P(ss_out), P(mr), N, Cout, None), 'finalize')
    torch.cuda.synchronize()
    yh = y.cpu().double()
    e = float((yh - ref).norm() / ref.norm())
    emax = float((yh - ref).abs().max() / ref.abs().max())
    mean, var = ref.mean((2, 3, 4)), ref.var((2, 3, 4), unbiased=False)
    em = float((mr[..., 0].cpu().double() - mean).abs().max())
    er = float((mr[..., 1].cpu().double() * torch.sqrt(var + 1e-5) - 1).abs().max())
    print('whole stage: rel L2 %.3g, max err / max|ref| %.3g; InstanceNorm mean err %.3g, rstd rel err %.3g' % (e, emax, em, er))
    assert e <= 2e-6 and emax <= 1e-5 and em <= 2e-5 and er <= 2e-5


def test_network_with_composed_stage_matches_the_two_kernel_form_and_the_oracle():
    """A network whose last decoder stage qualifies for the composed kernel (32 feature maps at 32^3): logits with and without it, and against
    the CPU oracle -- the engine's wiring of the stage (skip-half image, composed image, moments of the sum feeding the next layer and the head)."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import tiny_plan
    from oracle.generic_unet import perturb_like_trained, state_dict_numpy, unet_from_plan
    torch.manual_seed(0)
    plan = tiny_plan(patch=32, base=32, num_pool=2, max_feat=128)
    net = perturb_like_trained(unet_from_plan(plan), seed=5).eval()
    net.do_ds = False
    sd = state_dict_numpy(net)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 1, 32, 32, 32, generator=g)
    with torch.no_grad():
        ref = net(x).numpy()
    out = {}
    for fused in (False, True):
        eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2', fuse_up=fused)
        out[fused] = eng.forward_logits(x.cuda()).cpu().numpy()
        took = [k for k in eng._bufs if k.startswith('tu')]
        # the last stage's transposed convolution ran (its buffer exists) only in the two-kernel form
        assert ('tu%d.y' % (len(eng.dec) - 1) in took) == (not fused), took
    scale = float(np.abs(ref).max())
    d = float(np.abs(out[True] - out[False]).max()) / scale
    e = float(np.abs(out[True] - ref).max()) / scale
    print('composed vs two-kernel stage: max |d logits| / max |logits| %.3g; composed vs the oracle %.3g' % (d, e))
    assert d <= 2e-5 and e <= 5e-5
