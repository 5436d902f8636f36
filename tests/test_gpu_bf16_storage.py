"""NCDHW bf16 TENSORS at the full-resolution level (round 1's storage mode, now storage='bf16_ncdhw'; storage='bf16' is the
channels-last form, tests/test_gpu_cl.py), through the C ABI.

Every "_st" call is the plain call with a format flag per side, so the bar is bit-exactness against the plain call:
  y_bf16: the stored tensor equals the fp32 result rounded to bf16 (round to nearest even = torch's cast), the
          InstanceNorm partial records are identical (they come from the fp32 accumulators);
  x_bf16: the result equals the plain call on the widened (bf16 -> fp32) sources, bit for bit.
The engine-level test then bounds the end-to-end effect on the softmax against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SLOPE = 0.01


@pytest.fixture(scope='module')
def lib():
    from deepwmh_amd import _lib
    lib = _lib.load()
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    _lib.check(lib.wmh_device_check(0), 'wmh_device_check')
    return lib


def P(t):
    from deepwmh_amd import _lib
    return _lib.ptr(t)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ss(g, N, Cc):
    return torch.stack([0.5 + torch.rand(N, Cc, generator=g), 0.3 * torch.randn(N, Cc, generator=g)], -1)


def _bits(t):
    return t.view(torch.int16).cpu().numpy() if t.dtype == torch.bfloat16 else t.view(torch.int32).cpu().numpy()


BF16_ST_CASES = [
    # N, C0, C1, Cout, D, H, W, ss0, ss1
    (1, 32, 0, 32, 8, 16, 64, True, False),
    (2, 32, 32, 32, 8, 8, 32, False, True),       # decoder concatenation: up-sampled tensor + skip
    (1, 6, 0, 40, 5, 9, 36, True, False),         # ragged tiles, cout past one row tile, channels past the chunk
    (1, 16, 16, 24, 6, 20, 44, True, True),
]


@pytest.mark.parametrize('case', BF16_ST_CASES, ids=[str(c) for c in BF16_ST_CASES])
@pytest.mark.parametrize('fmt', [(0, 1), (1, 0), (1, 1)], ids=['out16', 'in16', 'in16+out16'])
def test_conv_bf16_operands_with_bf16_tensors(lib, case, fmt):
    from deepwmh_amd import _lib
    N, C0, C1, Cout, D, H, W, use0, use1 = case
    in16, out16 = fmt
    g = torch.Generator().manual_seed(5)
    Cin = C0 + C1
    x0 = (torch.randn(N, C0, D, H, W, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    x1 = torch.randn(N, C1, D, H, W, generator=g).to(torch.bfloat16) if C1 else None
    ss0 = _ss(g, N, C0) if use0 else None
    ss1 = _ss(g, N, C1) if (use1 and C1) else None
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1)
    assert cfg == 0
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cin, Cout), np.uint16)
    assert lib.wmh_conv3d_k3_pack_weights_bf16(cfg, w.numpy().ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, D, H, W)
    wfd, bd = dev(wf.view(np.int16)), b.cuda()
    ssd = [t.cuda() if t is not None else None for t in (ss0, ss1)]
    # plain call on the widened sources
    xf = [x0.float().cuda(), x1.float().cuda() if C1 else None]
    y_ref = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
    p_ref = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(xf[0]), P(ssd[0]), C0, P(xf[1]), P(ssd[1]), C1, P(wfd), P(bd), P(y_ref), P(p_ref),
                                          N, D, H, W, Cout, SLOPE, None), 'plain')
    xs = [x0.cuda(), x1.cuda() if C1 else None] if in16 else xf
    y = torch.full((N, Cout, D, H, W), float('nan'), device='cuda', dtype=torch.bfloat16 if out16 else torch.float32)
    p = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd_bf16_st(cfg, P(xs[0]), P(ssd[0]), C0, P(xs[1]), P(ssd[1]), C1, P(wfd), P(bd), P(y), P(p),
                                             N, D, H, W, Cout, SLOPE, in16, out16, None), 'st')
    torch.cuda.synchronize()
    assert not torch.isnan(y_ref).any()
    want = y_ref.to(torch.bfloat16) if out16 else y_ref
    assert np.array_equal(_bits(y), _bits(want))
    assert np.array_equal(_bits(p), _bits(p_ref))


def test_bf16_storage_rejects_unserved_combinations(lib):
    x = torch.zeros(1, 16, 8, 8, 16, device='cuda')                      # 16-wide rows: config 2
    cfg = lib.wmh_conv3d_k3_config(16, 16, 8, 8, 16, 1)
    assert cfg == 2
    wf = torch.zeros(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, 16, 16), dtype=torch.int16, device='cuda')
    y = torch.zeros(1, 16, 8, 8, 16, device='cuda')
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, 1, 8, 8, 16)
    part = torch.zeros(1, 16, npart, 4, device='cuda')
    assert lib.wmh_conv3d_k3_fwd_bf16_st(cfg, P(x), None, 16, None, None, 0, P(wf), None, P(y), P(part), 1, 8, 8, 16, 16, SLOPE, 0, 1, None) != 0
    wf32 = torch.zeros(lib.wmh_conv3d_k3_packed_floats(cfg, 16, 16), device='cuda')
    assert lib.wmh_conv3d_k3_fwd_st(cfg, P(x), None, 16, None, None, 0, P(wf32), None, P(y), P(part), 1, 8, 8, 16, 16, 1, SLOPE, 1, 0, None) != 0
    assert lib.wmh_conv3d_k3_fwd_st(cfg, P(x), None, 16, None, None, 0, P(wf32), None, P(y), P(part), 1, 8, 8, 16, 16, 1, SLOPE, 0, 1, None) != 0


@pytest.mark.parametrize('case', [(1, 1, 32, 16, 16, 32), (2, 2, 40, 12, 16, 40), (1, 1, 8, 9, 17, 33)], ids=str)
def test_first_layer_writes_bf16(lib, case):
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W = case
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, Cin, D, H, W, generator=g).cuda()
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.3
    b = (0.1 * torch.randn(Cout, generator=g)).cuda()
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1)
    assert cfg == 8
    wf = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout), np.float32)
    assert lib.wmh_conv3d_k3_pack_weights(cfg, w.numpy().ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    wfd = dev(wf)
    npart = lib.wmh_conv3d_k3_num_partials(cfg, N, D, H, W)
    y_ref = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
    p_ref = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, P(x), None, Cin, None, None, 0, P(wfd), P(b), P(y_ref), P(p_ref), None,
                                     N, D, H, W, Cout, 1, SLOPE, None), 'plain')
    y = torch.full((N, Cout, D, H, W), float('nan'), device='cuda', dtype=torch.bfloat16)
    p = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd_st(cfg, P(x), None, Cin, None, None, 0, P(wfd), P(b), P(y), P(p),
                                        N, D, H, W, Cout, 1, SLOPE, 0, 1, None), 'st')
    torch.cuda.synchronize()
    assert not torch.isnan(y_ref).any()
    assert np.array_equal(_bits(y), _bits(y_ref.to(torch.bfloat16)))
    assert np.array_equal(_bits(p), _bits(p_ref))


@pytest.mark.parametrize('case', [(1, 32, 64, 16, 32, 64), (2, 8, 16, 32, 32, 64), (1, 5, 8, 9, 11, 72)], ids=str)
def test_first_strided_convolution_reads_bf16(lib, case):
    from deepwmh_amd import _lib
    N, Cin, Cout, D, H, W = case
    g = torch.Generator().manual_seed(7)
    x16 = (torch.randn(N, Cin, D, H, W, generator=g) * 1.5).to(torch.bfloat16).cuda()
    xf = x16.float()
    ss = _ss(g, N, Cin).cuda()
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b = (0.1 * torch.randn(Cout, generator=g)).cuda()
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 2)
    assert cfg == 1
    wf = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout), np.float32)
    assert lib.wmh_conv3d_k3_pack_weights(cfg, w.numpy().ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)) == 0
    wfd = dev(wf)
    Do, Ho, Wo = (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1
    npart = lib.wmh_conv3d_k3_num_partials(cfg, N, Do, Ho, Wo)
    y_ref = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    p_ref = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    # no workspace: unsplit, like the _st call
    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, P(xf), P(ss), Cin, None, None, 0, P(wfd), P(b), P(y_ref), P(p_ref), None,
                                     N, D, H, W, Cout, 2, SLOPE, None), 'plain')
    y = torch.full((N, Cout, Do, Ho, Wo), float('nan'), device='cuda')
    p = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd_st(cfg, P(x16), P(ss), Cin, None, None, 0, P(wfd), P(b), P(y), P(p),
                                        N, D, H, W, Cout, 2, SLOPE, 1, 0, None), 'st')
    torch.cuda.synchronize()
    assert not torch.isnan(y_ref).any()
    assert np.array_equal(_bits(y), _bits(y_ref))
    assert np.array_equal(_bits(p), _bits(p_ref))


@pytest.mark.parametrize('shape', [(2, 16, 8, 2, 4, 4, 4), (1, 64, 32, 2, 8, 8, 16), (2, 24, 20, 1, 3, 8, 8)], ids=str)
def test_convtranspose_writes_bf16(lib, shape):
    from deepwmh_amd import _lib
    N, Cin, Cout, kt, D, H, W = shape
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, Cin, D, H, W, generator=g).cuda()
    ss = _ss(g, N, Cin).cuda()
    w = torch.randn(Cin, Cout, kt, 2, 2, generator=g) * 0.2
    wf = np.empty(lib.wmh_convtranspose3d_packed_floats(Cin, Cout, kt), np.float32)
    assert lib.wmh_convtranspose3d_pack_weights(w.numpy().ctypes.data_as(C.c_void_p), Cin, Cout, kt, wf.ctypes.data_as(C.c_void_p)) == 0
    wfd = dev(wf)
    y_ref = torch.full((N, Cout, kt * D, 2 * H, 2 * W), float('nan'), device='cuda')
    _lib.check(lib.wmh_convtranspose3d_mfma_fwd(P(x), P(ss), P(wfd), P(y_ref), N, Cin, Cout, D, H, W, kt, SLOPE, None), 'plain')
    y = torch.full(y_ref.shape, float('nan'), device='cuda', dtype=torch.bfloat16)
    _lib.check(lib.wmh_convtranspose3d_mfma_fwd_st(P(x), P(ss), P(wfd), P(y), N, Cin, Cout, D, H, W, kt, SLOPE, 1, None), 'st')
    y32 = torch.full(y_ref.shape, float('nan'), device='cuda')
    _lib.check(lib.wmh_convtranspose3d_mfma_fwd_st(P(x), P(ss), P(wfd), P(y32), N, Cin, Cout, D, H, W, kt, SLOPE, 0, None), 'st fp32')
    torch.cuda.synchronize()
    assert not torch.isnan(y_ref).any()
    assert np.array_equal(_bits(y), _bits(y_ref.to(torch.bfloat16)))
    assert np.array_equal(_bits(y32), _bits(y_ref))


@pytest.mark.parametrize('P2', [16, 10], ids=['quads', 'scalar'])
def test_head_reads_bf16(lib, P2):
    from deepwmh_amd import _lib
    g = torch.Generator().manual_seed(9)
    n, Cc, K, P0, P1 = 8, 32, 2, 6, 8
    x16 = torch.randn(n, Cc, P0, P1, P2, generator=g).to(torch.bfloat16).cuda()
    xf = x16.float()
    ss = _ss(g, n, Cc).cuda()
    wseg = (0.3 * torch.randn(K, Cc, generator=g)).cuda()
    gauss = (0.1 + torch.rand(P0, P1, P2, generator=g)).cuda()
    A = (P0 + 2, P1 + 3, P2 + 4)
    flips = list(range(8))
    outs = []
    for x, f in ((xf, 0), (x16, 1)):
        agg = torch.zeros((K,) + A, device='cuda')
        _lib.check(lib.wmh_seghead_softmax_accumulate_st(P(x), P(ss), P(wseg), K, P(gauss), P(agg), Cc, P0, P1, P2, A[0], A[1], A[2],
                                                         _lib.iarr([1, 2, 4 if P2 % 4 == 0 else 3]), n, _lib.iarr(flips), SLOPE, f, None), 'head')
        lg = torch.full((n, K, P0, P1, P2), float('nan'), device='cuda')
        _lib.check(lib.wmh_seghead_logits_st(P(x), P(ss), P(wseg), K, P(lg), n, Cc, P0 * P1 * P2, SLOPE, f, None), 'logits')
        outs.append((agg, lg))
    torch.cuda.synchronize()
    assert float(outs[0][0].abs().sum()) > 0 and not torch.isnan(outs[0][1]).any()
    assert np.array_equal(_bits(outs[0][0]), _bits(outs[1][0]))
    assert np.array_equal(_bits(outs[0][1]), _bits(outs[1][1]))


def test_engine_with_bf16_storage_close_to_oracle():
    """End to end: the bf16-operand engine with bf16 full-resolution tensors against the fp32 CPU oracle and against the
    same engine with fp32 tensors, logits and sliding-window prediction with mirroring.  Bounds as for the bf16-operand
    configuration (BASELINE.json config 3): softmax within 5e-2 of fp32, Dice against the fp32 mask > 0.98."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import UNetPlan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    torch.manual_seed(4)
    plan = UNetPlan(1, 16, 2, [[2, 2, 2]] * 2, [[3, 3, 3]] * 3, 2, 64, [16, 32, 48])
    net = perturb_like_trained(unet_from_plan(plan), seed=6).eval()
    sd = state_dict_numpy(net)
    vol = osw.synthetic_flair((20, 40, 64), seed=10)
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, 2:18, 4:36, 8:56]))
    with torch.no_grad():
        net.do_ds = False
        ref = torch.softmax(net(x), 1)
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True)
    res = {}
    for storage in ('fp32', 'bf16_ncdhw'):
        eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage=storage)
        feats = eng.forward_features(x.cuda())
        assert feats[-1][0].dtype == (torch.bfloat16 if storage == 'bf16_ncdhw' else torch.float32)
        sm = torch.softmax(eng.forward_logits(x.cuda()), 1).cpu()
        seg, prob = SlidingWindowPredictor(eng, do_mirroring=True).predict(vol)
        res[storage] = (sm, seg, prob)
    d_or = float((res['bf16_ncdhw'][0] - ref).abs().max())
    d_st = float((res['bf16_ncdhw'][0] - res['fp32'][0]).abs().max())
    p_or = float(np.abs(res['bf16_ncdhw'][2] - prob_o).max())
    dice = osw.hard_dice_binary(seg_o, res['bf16_ncdhw'][1])
    print('bf16 storage: softmax max abs vs oracle %.3g, vs fp32 storage %.3g; tiled+mirrored vs oracle %.3g, Dice %.5f' % (d_or, d_st, p_or, dice))
    assert d_or < 5e-2 and d_st < 5e-2 and p_or < 5e-2 and dice > 0.98
    with pytest.raises(ValueError):
        UNetEngine(plan, sd, 'cuda:0', precision='fp32', storage='bf16_ncdhw')


def test_storage_mode_is_refused_up_front_for_plans_it_does_not_serve(tiny_net):
    from deepwmh_amd.network import UNetEngine
    plan, _, sd = tiny_net                          # 4 feature maps at full resolution, 32-voxel rows
    with pytest.raises(ValueError, match='not available for this plan'):
        UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage='bf16_ncdhw')
