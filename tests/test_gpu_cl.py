"""Channels-last bf16 kernels (csrc/cl_*.hip; storage='bf16' of BASELINE.json configs[2]) through the C ABI, against
torch on the CPU.  Reference arithmetic = what the kernels are specified to do: sources are bf16 values, the pending
InstanceNorm + LeakyReLU is applied in fp32 (one fused multiply-add), the result is rounded to bf16 (nearest even),
multiplied with bf16-rounded weights, accumulated in fp32 (reference: float64), bias added, output rounded to bf16;
InstanceNorm moments come from the un-rounded accumulators.  Tolerance: one bf16 rounding of the output (2^-8 relative)
plus the fp32 accumulation-order term."""
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


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def to_cl(x):
    """numpy [N,C,D,H,W] (values already bf16-representable or not: rounded) -> cuda bf16 tensor in the CL16 format of cl_common.h,
    [N, C/8, D, H, W, 8]: planes of 8 channels, the 8 channels of a voxel contiguous."""
    N, C, D, H, W = x.shape
    assert C % 8 == 0
    t = np.ascontiguousarray(x.reshape(N, C // 8, 8, D, H, W).transpose(0, 1, 3, 4, 5, 2))
    return torch.from_numpy(t).to(torch.bfloat16).cuda().contiguous()


def from_cl(t):
    """CL16 tensor -> numpy [N,C,D,H,W] fp32.  A 5-D tensor [N, D, H, W, C] is taken as a buffer of that many elements handed to a
    kernel as its output (the tests allocate them that way): the memory is [N, C/8, D, H, W, 8]."""
    if t.dim() == 5:
        N, D, H, W, C = t.shape
        t = t.contiguous().reshape(N, C // 8, D, H, W, 8)
    N, C8, D, H, W, _ = t.shape
    return np.ascontiguousarray(t.float().cpu().numpy().transpose(0, 1, 5, 2, 3, 4)).reshape(N, C8 * 8, D, H, W)


def pending(xq, ss):
    """lrelu(fma(x, scale, shift)) in fp32, rounded to bf16 -- the operand the matrix cores see."""
    if ss is None:
        return xq
    sc = ss[:, :, 0][:, :, None, None, None].astype(np.float64)
    sh = ss[:, :, 1][:, :, None, None, None].astype(np.float64)
    z = (xq.astype(np.float64) * sc + sh).astype(np.float32)
    z = np.maximum(z, (z * np.float32(SLOPE)).astype(np.float32))
    return bf16_round(z)


def close(got, ref, what, rel=2.0 ** -7, abs_=None):
    abs_ = 2e-3 * float(np.abs(ref).max()) if abs_ is None else abs_
    err = np.abs(got - ref)
    bad = err > rel * np.abs(ref) + abs_
    assert not bad.any(), '%s: %d of %d outside tolerance, max err %.4g (|ref| max %.4g)' % (what, int(bad.sum()), bad.size, float(err.max()), float(np.abs(ref).max()))
    return float(err.max())


def finalize(lib, part, npart, N, Cout):
    from deepwmh_amd import _lib
    ss = torch.empty(N, Cout, 2, device='cuda')
    g = torch.ones(Cout, device='cuda')
    b = torch.zeros(Cout, device='cuda')
    _lib.check(lib.wmh_instnorm_finalize(_lib.ptr(part), npart, _lib.ptr(g), _lib.ptr(b), 1e-5, _lib.ptr(ss), None, N, Cout, None), 'finalize')
    return ss.cpu().numpy()


def ref_scale_shift(y64):
    m = y64.mean(axis=(2, 3, 4))
    v = y64.var(axis=(2, 3, 4))
    sc = 1.0 / np.sqrt(v + 1e-5)
    return np.stack([sc, -m * sc], -1)


def pack_w(lib, w):
    from deepwmh_amd import _lib
    cout, cin = w.shape[:2]
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(0, cin, cout), dtype=np.uint16)
    w = np.ascontiguousarray(w, dtype=np.float32)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_bf16(0, w.ctypes.data_as(C.c_void_p), cin, cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    return torch.from_numpy(wf.view(np.int16)).cuda()


def test_layout_converters_round_trip(lib):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 24, 3, 5, 7)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    y = torch.empty(2, 3, 5, 7, 24, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.wmh_cl_from_ncdhw(_lib.ptr(xd), _lib.ptr(y), 2, 24, 3 * 5 * 7, None), 'from')
    assert np.array_equal(from_cl(y), bf16_round(x))
    back = torch.full((2, 24, 3, 5, 7), float('nan'), device='cuda')
    _lib.check(lib.wmh_cl_to_ncdhw(_lib.ptr(y), _lib.ptr(back), 2, 24, 3 * 5 * 7, None), 'to')
    assert np.array_equal(back.cpu().numpy(), bf16_round(x))
    assert lib.wmh_cl_from_ncdhw(_lib.ptr(xd), _lib.ptr(y), 2, 12, 105, None) < 0        # C % 8


@pytest.mark.parametrize('N,D,H,W,Cout', [(2, 8, 8, 32, 32), (1, 5, 11, 64, 64), (1, 4, 8, 32, 32)])
def test_first_convolution(lib, N, D, H, W, Cout):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(1)
    x = rng.standard_normal((N, 1, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, 1, 3, 3, 3)) / np.sqrt(27)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    wf = np.empty(lib.wmh_cl_conv3_first_packed_halfs(Cout), dtype=np.uint16)
    _lib.check(lib.wmh_cl_conv3_first_pack_weights(w.ctypes.data_as(C.c_void_p), Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    npart = lib.wmh_cl_conv3_first_num_partials(D, H, W)
    y = torch.full((N, D, H, W, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    xd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(bias).cuda()
    _lib.check(lib.wmh_cl_conv3_first_fwd(_lib.ptr(xd), _lib.ptr(wfd), _lib.ptr(bd), _lib.ptr(y), _lib.ptr(part), N, D, H, W, Cout, None), 'first')
    ref = F.conv3d(torch.from_numpy(bf16_round(x)).double(), torch.from_numpy(bf16_round(w)).double(), torch.from_numpy(bias).double(), padding=1).numpy()
    close(from_cl(y), ref, 'first conv')
    assert not torch.isnan(part).any()
    close(finalize(lib, part, npart, N, Cout), ref_scale_shift(ref), 'first conv scale/shift', rel=2e-4, abs_=2e-4)


CASES = [
    # N, D, H, W, C0, C1, Cout, ss0, ss1
    (1, 4, 4, 32, 32, 0, 32, True, False),
    (2, 8, 8, 32, 16, 0, 32, True, False),
    (1, 6, 7, 64, 32, 32, 64, False, True),      # decoder form: up-sampled source without pending norm + skip with one; ragged D, H
    (1, 9, 5, 32, 48, 16, 32, True, True),
    (1, 4, 8, 96, 32, 0, 96, True, False),
]


@pytest.mark.parametrize('N,D,H,W,C0,C1,Cout,has0,has1', CASES)
def test_conv3_stride1(lib, N, D, H, W, C0, C1, Cout, has0, has1):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(2)
    Cin = C0 + C1
    x = bf16_round(rng.standard_normal((N, Cin, D, H, W)).astype(np.float32) * 1.5 + 0.3)
    ss = np.stack([1 + 0.3 * rng.standard_normal((N, Cin)), 0.4 * rng.standard_normal((N, Cin))], -1).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    x0, x1 = to_cl(x[:, :C0]), (to_cl(x[:, C0:]) if C1 else None)
    ss0 = torch.from_numpy(np.ascontiguousarray(ss[:, :C0])).cuda() if has0 else None
    ss1 = torch.from_numpy(np.ascontiguousarray(ss[:, C0:])).cuda() if (C1 and has1) else None
    npart = lib.wmh_cl_conv3_num_partials(D, H, W)
    y = torch.full((N, D, H, W, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    bd = torch.from_numpy(bias).cuda()
    wfd = pack_w(lib, w)
    _lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wfd), _lib.ptr(bd),
                                    _lib.ptr(y), _lib.ptr(part), N, D, H, W, Cout, SLOPE, None), 'cl_conv3')
    torch.cuda.synchronize()
    xin = np.concatenate([pending(x[:, :C0], ss[:, :C0] if has0 else None)] + ([pending(x[:, C0:], ss[:, C0:] if has1 else None)] if C1 else []), 1)
    ref = F.conv3d(torch.from_numpy(xin).double(), torch.from_numpy(bf16_round(w)).double(), torch.from_numpy(bias).double(), padding=1).numpy()
    got = from_cl(y)
    assert not np.isnan(got).any(), 'output voxels never written: %d' % int(np.isnan(got).sum())
    e = close(got, ref, 'cl_conv3')
    assert not torch.isnan(part).any()
    close(finalize(lib, part, npart, N, Cout), ref_scale_shift(ref), 'cl_conv3 scale/shift', rel=3e-4, abs_=3e-4)
    print('cl_conv3 %s: max err %.3g' % ((N, D, H, W, C0, C1, Cout), e))


def test_conv3_many_tiles_per_workgroup(lib):
    """More tiles than workgroups (the persistent loop: buffer ring, table slots, running moments across tiles and across a
    sample change inside a workgroup's run), two sources, two cout groups."""
    from deepwmh_amd import _lib
    rng = np.random.default_rng(7)
    N, D, H, W, C0, C1, Cout = 2, 40, 48, 64, 16, 16, 64          # 2 * 10 * 12 * 2 tiles * 2 groups = 960 tiles
    Cin = C0 + C1
    x = bf16_round(rng.standard_normal((N, Cin, D, H, W)).astype(np.float32) * 1.5 + 0.3)
    ss = np.stack([1 + 0.3 * rng.standard_normal((N, Cin)), 0.4 * rng.standard_normal((N, Cin))], -1).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    x0, x1 = to_cl(x[:, :C0]), to_cl(x[:, C0:])
    ss0 = torch.from_numpy(np.ascontiguousarray(ss[:, :C0])).cuda()
    ss1 = torch.from_numpy(np.ascontiguousarray(ss[:, C0:])).cuda()
    npart = lib.wmh_cl_conv3_num_partials(D, H, W)
    y = torch.full((N, D, H, W, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    bd, wfd = torch.from_numpy(bias).cuda(), pack_w(lib, w)
    _lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wfd), _lib.ptr(bd),
                                    _lib.ptr(y), _lib.ptr(part), N, D, H, W, Cout, SLOPE, None), 'cl_conv3')
    torch.cuda.synchronize()
    xin = np.concatenate([pending(x[:, :C0], ss[:, :C0]), pending(x[:, C0:], ss[:, C0:])], 1)
    ref = F.conv3d(torch.from_numpy(xin).double(), torch.from_numpy(bf16_round(w)).double(), torch.from_numpy(bias).double(), padding=1).numpy()
    got = from_cl(y)
    assert not np.isnan(got).any(), 'output voxels never written: %d' % int(np.isnan(got).sum())
    close(got, ref, 'cl_conv3, many tiles')
    assert not torch.isnan(part).any()
    close(finalize(lib, part, npart, N, Cout), ref_scale_shift(ref), 'cl_conv3 scale/shift, many tiles', rel=3e-4, abs_=3e-4)


def test_conv3_is_deterministic_and_batch_independent(lib):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(3)
    x = bf16_round(rng.standard_normal((3, 32, 8, 8, 32)).astype(np.float32))
    ss = torch.from_numpy(np.stack([1 + 0.2 * rng.standard_normal((3, 32)), 0.3 * rng.standard_normal((3, 32))], -1).astype(np.float32)).cuda()
    w = (rng.standard_normal((32, 32, 3, 3, 3)) / 30).astype(np.float32)
    wf = pack_w(lib, w)
    xs = to_cl(x)
    npart = lib.wmh_cl_conv3_num_partials(8, 8, 32)

    def run(xc, sc, n):
        y = torch.empty((n, 8, 8, 32, 32), dtype=torch.bfloat16, device='cuda')
        part = torch.empty((n, 32, npart, 4), device='cuda')
        _lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(xc), _lib.ptr(sc), 32, None, None, 0, _lib.ptr(wf), None, _lib.ptr(y), _lib.ptr(part),
                                        n, 8, 8, 32, 32, SLOPE, None), 'cl_conv3')
        return y.view(torch.int16).cpu().numpy(), part.cpu().numpy(), finalize(lib, part, npart, n, 32)
    y3, p3, s3 = run(xs, ss, 3)
    y3b, p3b, _ = run(xs, ss, 3)
    assert np.array_equal(y3, y3b) and np.array_equal(p3, p3b)
    for b in range(3):
        # outputs do not depend on the batch; the moment records are cut differently (workgroups x samples), their sum is not
        y1, p1, s1 = run(xs[b:b + 1].contiguous(), ss[b:b + 1].contiguous(), 1)
        assert np.array_equal(y1[0], y3[b])
        assert np.allclose(s1[0], s3[b], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('N,D,H,W,Cin,Cout,has_ss', [(1, 4, 4, 64, 32, 64, True), (2, 8, 6, 64, 16, 32, True), (1, 5, 7, 128, 32, 64, False)])
def test_conv3_stride2(lib, N, D, H, W, Cin, Cout, has_ss):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(4)
    x = bf16_round(rng.standard_normal((N, Cin, D, H, W)).astype(np.float32) * 1.5 + 0.3)
    ss = np.stack([1 + 0.3 * rng.standard_normal((N, Cin)), 0.4 * rng.standard_normal((N, Cin))], -1).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, W // 2
    npart = lib.wmh_cl_conv3s2_num_partials(Do, Ho, Wo)
    y = torch.full((N, Do, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    part = torch.full((N, Cout, npart, 4), float('nan'), device='cuda')
    ssd = torch.from_numpy(ss).cuda() if has_ss else None
    bd = torch.from_numpy(bias).cuda()
    xc, wfd = to_cl(x), pack_w(lib, w)           # named: a temporary would be freed (and its memory reused) before the launch
    _lib.check(lib.wmh_cl_conv3s2_fwd(_lib.ptr(xc), _lib.ptr(ssd), Cin, _lib.ptr(wfd), _lib.ptr(bd), _lib.ptr(y), _lib.ptr(part),
                                      N, D, H, W, Cout, SLOPE, None), 'cl_conv3s2')
    torch.cuda.synchronize()
    xin = pending(x, ss if has_ss else None)
    ref = F.conv3d(torch.from_numpy(xin).double(), torch.from_numpy(bf16_round(w)).double(), torch.from_numpy(bias).double(), stride=2, padding=1).numpy()
    got = from_cl(y)
    assert got.shape == ref.shape and not np.isnan(got).any()
    close(got, ref, 'cl_conv3s2')
    assert not torch.isnan(part).any()
    close(finalize(lib, part, npart, N, Cout), ref_scale_shift(ref), 'cl_conv3s2 scale/shift', rel=3e-4, abs_=3e-4)


@pytest.mark.parametrize('N,D,H,W,Cin,Cout,has_ss', [(1, 2, 4, 32, 64, 32, True), (2, 4, 4, 16, 32, 64, True), (1, 2, 2, 8, 16, 32, False)])
def test_transposed_convolution(lib, N, D, H, W, Cin, Cout, has_ss):
    from deepwmh_amd import _lib
    rng = np.random.default_rng(5)
    x = bf16_round(rng.standard_normal((N, Cin, D, H, W)).astype(np.float32))
    ss = np.stack([1 + 0.3 * rng.standard_normal((N, Cin)), 0.4 * rng.standard_normal((N, Cin))], -1).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, 2, 2, 2)) / np.sqrt(Cin)).astype(np.float32)
    wf = np.empty(lib.wmh_cl_convt_packed_halfs(Cin, Cout), dtype=np.uint16)
    _lib.check(lib.wmh_cl_convt_pack_weights(w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    y = torch.full((N, 2 * D, 2 * H, 2 * W, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    ssd = torch.from_numpy(ss).cuda() if has_ss else None
    xc, wfd = to_cl(x), torch.from_numpy(wf.view(np.int16)).cuda()
    _lib.check(lib.wmh_cl_convt_fwd(_lib.ptr(xc), _lib.ptr(ssd), _lib.ptr(wfd), _lib.ptr(y),
                                    N, Cin, Cout, D, H, W, SLOPE, None), 'cl_convT')
    torch.cuda.synchronize()
    ref = F.conv_transpose3d(torch.from_numpy(pending(x, ss if has_ss else None)).double(), torch.from_numpy(bf16_round(w)).double(), stride=2).numpy()
    got = from_cl(y)
    assert not np.isnan(got).any()
    close(got, ref, 'cl_convT')


def test_segmentation_head_matches_the_ncdhw_kernel_bit_for_bit(lib):
    """Same arithmetic and order as wmh_seghead_softmax_accumulate on the same (bf16) feature map: equal bits."""
    from deepwmh_amd import _lib
    rng = np.random.default_rng(6)
    P, A, org = (8, 12, 16), (10, 16, 20), (1, 3, 2)
    n, Cc, K = 8, 32, 2
    x = bf16_round(rng.standard_normal((n, Cc) + P).astype(np.float32))
    ss = torch.from_numpy(np.stack([1 + 0.3 * rng.standard_normal((n, Cc)), 0.4 * rng.standard_normal((n, Cc))], -1).astype(np.float32)).cuda()
    wseg = torch.from_numpy(rng.standard_normal((K, Cc)).astype(np.float32)).cuda()
    gauss = torch.from_numpy(rng.uniform(0.1, 1, P).astype(np.float32)).cuda()
    base = rng.standard_normal((K,) + A).astype(np.float32)
    flips = list(range(8))
    outs = []
    for cl in (False, True):
        agg = torch.from_numpy(base).cuda()
        if cl:
            xc = to_cl(x)
            _lib.check(lib.wmh_cl_seghead_softmax_accumulate(_lib.ptr(xc), _lib.ptr(ss), _lib.ptr(wseg), K, _lib.ptr(gauss), _lib.ptr(agg), Cc,
                                                             P[0], P[1], P[2], A[0], A[1], A[2], _lib.iarr(org), n, _lib.iarr(flips), SLOPE, None), 'cl head')
        else:
            xd = torch.from_numpy(x).cuda()
            _lib.check(lib.wmh_seghead_softmax_accumulate(_lib.ptr(xd), _lib.ptr(ss), _lib.ptr(wseg), K, _lib.ptr(gauss), _lib.ptr(agg), Cc,
                                                          P[0], P[1], P[2], A[0], A[1], A[2], _lib.iarr(org), n, _lib.iarr(flips), SLOPE, None), 'head')
        outs.append(agg.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], base)


def test_argument_checks(lib):
    from deepwmh_amd import _lib
    t = torch.zeros(64, device='cuda')
    assert lib.wmh_cl_conv3_fwd(_lib.ptr(t), None, 24, None, None, 0, _lib.ptr(t), None, _lib.ptr(t), None, 1, 4, 4, 32, 32, SLOPE, None) < 0   # C0 % 16
    assert lib.wmh_cl_conv3_fwd(_lib.ptr(t), None, 32, None, None, 0, _lib.ptr(t), None, _lib.ptr(t), None, 1, 4, 4, 48, 32, SLOPE, None) < 0   # W % 32
    assert lib.wmh_cl_conv3s2_fwd(_lib.ptr(t), None, 32, _lib.ptr(t), None, _lib.ptr(t), None, 1, 4, 4, 32, 32, SLOPE, None) < 0              # W % 64
    assert lib.wmh_cl_convt_fwd(_lib.ptr(t), None, _lib.ptr(t), _lib.ptr(t), 1, 32, 32, 1, 3, 5, SLOPE, None) < 0                             # voxels % 32
    assert lib.wmh_cl_conv3_num_partials(8, 8, 40) < 0


def _cl_plan():
    from deepwmh_amd.plan import UNetPlan
    return UNetPlan(1, 32, 2, [[2, 2, 2]] * 3, [[3, 3, 3]] * 4, 2, 128, [16, 32, 64])


def test_engine_channels_last_close_to_oracle():
    """End to end: two channels-last levels (rows of 64 and 32 voxels) over two NCDHW levels, against the fp32 CPU oracle and
    against the same engine with fp32 tensors; logits and the tiled + mirrored prediction.  Bounds as for the bf16-operand
    configuration (BASELINE.json configs[2]): softmax within 5e-2 of fp32, Dice against the fp32 mask > 0.98."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    torch.manual_seed(4)
    plan = _cl_plan()
    net = perturb_like_trained(unet_from_plan(plan), seed=6).eval()
    sd = state_dict_numpy(net)
    vol = osw.synthetic_flair((20, 40, 80), seed=10)
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, 2:18, 4:36, 8:72]))
    with torch.no_grad():
        net.do_ds = False
        ref = torch.softmax(net(x), 1)
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True)
    res = {}
    for storage in ('fp32', 'bf16'):
        eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage=storage)
        assert eng.ncl == (2 if storage == 'bf16' else 0)
        feats = eng.forward_features(x.cuda())
        assert feats[-1][0].dtype == (torch.bfloat16 if storage == 'bf16' else torch.float32)
        sm = torch.softmax(eng.forward_logits(x.cuda()), 1).cpu()
        seg, prob = SlidingWindowPredictor(eng, do_mirroring=True).predict(vol)
        seg_n, prob_n = SlidingWindowPredictor(eng, do_mirroring=False, batch=3).predict(vol)
        res[storage] = (sm, seg, prob, prob_n)
    d_or = float((res['bf16'][0] - ref).abs().max())
    d_st = float((res['bf16'][0] - res['fp32'][0]).abs().max())
    p_or = float(np.abs(res['bf16'][2] - prob_o).max())
    dice = osw.hard_dice_binary(seg_o, res['bf16'][1])
    _, prob_on = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, False, True)
    p_nt = float(np.abs(res['bf16'][3] - prob_on).max())
    print('channels-last bf16: softmax max abs vs oracle %.3g, vs fp32 tensors %.3g; tiled+mirrored vs oracle %.3g, Dice %.5f; no mirroring %.3g'
          % (d_or, d_st, p_or, dice, p_nt))
    assert d_or < 5e-2 and d_st < 5e-2 and p_or < 5e-2 and p_nt < 5e-2 and dice > 0.98


def test_storage_mode_is_refused_up_front_for_plans_it_does_not_serve(tiny_net):
    from deepwmh_amd.network import UNetEngine
    plan, _, sd = tiny_net                          # 4 feature maps at full resolution
    with pytest.raises(ValueError, match='not available for this plan'):
        UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage='bf16')
    with pytest.raises(ValueError):
        UNetEngine(_cl_plan(), {}, 'cuda:0', precision='fp32', storage='bf16')


def test_configs2_full_width_patch_against_the_fp32_path():
    """BASELINE.json configs[2] at its own size: the 31 M-parameter plan, a batch of 128^3 patches, channels-last bf16 tensors on
    the three wide levels against the exact-fp32 path (itself within 1.5e-5 of the oracle at this size, test_gpu_network.py).
    Random He-init weights keep the probabilities near 0.5 -- the worst case for mask agreement."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    vol = synthetic_flair((128, 128, 160), seed=3)
    x = torch.from_numpy(np.ascontiguousarray(np.stack([vol[:, :, :, :128], vol[:, :, :, 32:]]))).cuda()
    ref = torch.softmax(UNetEngine(plan, sd, 'cuda:0').forward_logits(x), 1)
    eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage='bf16')
    assert eng.ncl == 3
    a = torch.softmax(eng.forward_logits(x), 1)
    b = torch.softmax(eng.forward_logits(x), 1)
    assert torch.equal(a, b)                                     # no atomics, no race: bit-reproducible
    assert bool(torch.isfinite(a).all())
    d = float((a - ref).abs().max())
    ma, mr = a[:, 1] > 0.5, ref[:, 1] > 0.5
    dice = 2.0 * float((ma & mr).sum()) / max(1.0, float(ma.sum() + mr.sum()))
    flips = float((ma != mr).float().mean())
    mean_d = float((a - ref).abs().mean())
    print('configs[2] full width: |softmax - fp32| max %.3g mean %.3g, Dice vs fp32 mask %.5f, %.3f %% of voxels change class'
          % (d, mean_d, dice, 100 * flips))
    # one un-mirrored, un-aggregated forward (the bench's mirrored + Gaussian-aggregated volume agrees to 0.018 / Dice 0.9986);
    # measured here: max 0.095, Dice 0.9965, 0.5 % of voxels
    assert d < 0.15 and mean_d < 1e-2 and dice > 0.99 and flips < 1e-2


def test_configs2_stress_eight_volumes_three_repetitions():
    """BASELINE.json configs[2] at the size its LDS-DMA ordering race showed at (garbage tiles "only at full size",
    every small test green): 8 volumes of 160x192x160, each 8 tiles x 8 mirrors = 64 patches in forwards of 16 (two tiles' mirrored copies,
    as bench.py and nnUNet_predict run it; rounds 2-3 ran it in forwards of 8) of the 31 M-parameter net on channels-last bf16 tensors, three repetitions.  Every result finite, all three repetitions bit-equal, masks against the fp32
    path."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    vols = [torch.from_numpy(synthetic_flair((160, 192, 160), seed=v)).cuda() for v in range(8)]
    eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage='bf16')
    assert eng.ncl == 3
    pred = SlidingWindowPredictor(eng, do_mirroring=True, batch=16)      # what bench.py and nnUNet_predict use: two tiles' mirrored copies per forward
    first = None
    for rep in range(3):
        outs = []
        for v in vols:
            seg, prob = pred.predict_device(v, return_prob=True)
            outs.append((seg.clone(), prob.clone()))
        torch.cuda.synchronize()
        for i, (seg, prob) in enumerate(outs):
            assert bool(torch.isfinite(prob).all()), 'repetition %d volume %d: non-finite softmax' % (rep, i)
            assert float((prob.sum(0) - 1).abs().max()) <= 1e-5
        if first is None:
            first = outs
        else:
            for i, ((s0, p0), (s1, p1)) in enumerate(zip(first, outs)):
                assert torch.equal(p0, p1) and torch.equal(s0, s1), 'repetition %d volume %d differs from repetition 0' % (rep, i)
    del eng, pred
    eng32 = UNetEngine(plan, sd, 'cuda:0')
    pred32 = SlidingWindowPredictor(eng32, do_mirroring=True, batch=8)
    worst_dice, worst_diff = 1.0, 0.0
    for i in (0, 3, 7):
        s32, p32 = pred32.predict_device(vols[i], return_prob=True)
        s16, p16 = first[i]
        inter = float(((s16 > 0) & (s32 > 0)).sum())
        dice = 2 * inter / (float((s16 > 0).sum()) + float((s32 > 0).sum()) + 1e-6)
        worst_dice, worst_diff = min(worst_dice, dice), max(worst_diff, float((p16 - p32).abs().max()))
    print('configs[2] stress: 3 x 8 volumes bit-equal and finite; vs the fp32 path on volumes 0, 3, 7: worst Dice %.5f, max |softmax diff| %.3g'
          % (worst_dice, worst_diff))
    assert worst_dice > 0.99 and worst_diff < 5e-2


def test_configs2_full_volume_with_mirroring_against_the_oracle_fixtures():
    """configs[2] (channels-last bf16 tensors, bf16 matrix-core operands) at full size against the ORACLE, not only
    against the fp32 HIP path: the whole mirrored 160x192x160 volume on both 31 M-parameter fixture networks.  bf16 operands carry 8
    significant bits, so the bound is a |dp| / Dice bound, stated per network:
    max |softmax - oracle| <= 2e-2 on the fixture's lattice and margin band, mean error <= 1e-3, Dice >= 0.94 against the oracle's mask.
    Both fixture networks are random-init nets whose probabilities stay within a few 1e-2 of 0.5 with ~28 k foreground voxels of 4.9 M -- the
    worst case for mask agreement.  Measured: He init + trained-like perturbation (fullsize_tta.npz) 1.0e-2 / 2.1e-4 / Dice 0.949; plain He
    init (fullsize_tta_he.npz) 9.0e-3 / 2.0e-4 / Dice 0.952 (2 669 voxels change class).  (On the bench's network, whose foreground is 40x
    larger, the same mode keeps Dice 0.9986 against the fp32 path's mask.)"""
    import os
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from oracle.generic_unet import perturb_like_trained, state_dict_numpy, unet_from_plan
    from test_gpu_network import _fullsize_fixture_check, _state_abs_sum
    plan = baseline_plan()
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    for name, kind, perturbed, dice_min in (('fullsize_tta.npz', 'perturbed', True, 0.94), ('fullsize_tta_he.npz', 'He-init', False, 0.94)):
        torch.manual_seed(0)
        net = unet_from_plan(plan)
        if perturbed:
            net = perturb_like_trained(net, seed=2)
        net = net.eval()
        G = np.load(os.path.join(golden, name))
        assert abs(_state_abs_sum(net) - float(G['state_abs_sum'])) <= 1e-7 * float(G['state_abs_sum']), 'not the weights the fixture was made with'
        eng = UNetEngine(plan, state_dict_numpy(net), 'cuda:0', precision='bf16', storage='bf16')
        assert eng.ncl == 3
        err, mism, mean_err, dice = _fullsize_fixture_check(name, kind, [('configs[2]', eng)])['configs[2]']
        assert err <= 2e-2 and mean_err <= 1e-3 and dice >= dice_min, (name, err, mean_err, dice)
        del eng
        torch.cuda.empty_cache()
