"""f16x2 convolution (wmh_conv3d_k3_fwd_f16x2): error against a float64 reference next to the exact-fp32 kernel's, and timing at the
BASELINE shapes.  python tools/x3_check.py"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib  # noqa: E402
lib = _lib.load_diag() if (os.environ.get('WMH_STAMPS') or os.environ.get('WMH_DIAG_LIB')) else _lib.load()   # stamps and variant switches: diagnostic library only
P = _lib.ptr


def run(N, S, C0, C1, Cout, check=True, reps=5):
    Cin = C0 + C1
    g = torch.Generator(device='cuda').manual_seed(1)
    x0 = torch.randn(N, C0, S, S, S, device='cuda', generator=g)
    x1 = torch.randn(N, C1, S, S, S, device='cuda', generator=g) if C1 else None
    ss0 = torch.stack([1 + 0.3 * torch.randn(N, C0, device='cuda', generator=g), 0.2 * torch.randn(N, C0, device='cuda', generator=g)], -1).contiguous()
    ss1 = torch.stack([1 + 0.3 * torch.randn(N, C1, device='cuda', generator=g), 0.2 * torch.randn(N, C1, device='cuda', generator=g)], -1).contiguous() if C1 else None
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 27)) / np.sqrt(27 * Cin)).astype(np.float32)
    bias = torch.randn(Cout, device='cuda', generator=g) * 0.1
    cfg0 = lib.wmh_conv3d_k3_config(Cin, Cout, S, S, S, 1)
    cfg = lib.wmh_conv3d_k3_bf16_config(cfg0, Cout, S)
    assert lib.wmh_conv3d_k3_f16x2_supported(cfg), cfg
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), np.uint16)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    y = torch.empty(N, Cout, S, S, S, device='cuda')
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, S, S, S)
    part = torch.empty(N, Cout, npart, 4, device='cuda')

    def x3():
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wfd), P(bias), P(y), P(part), None, N, S, S, S, Cout, 0.01, None), 'x3')
    # the exact-fp32 kernel
    wf32 = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg0, Cin, Cout), np.float32)
    lib.wmh_conv3d_k3_pack_weights(cfg0, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf32.ctypes.data_as(C.c_void_p))
    wf32d = torch.from_numpy(wf32).cuda()
    y32 = torch.empty_like(y)
    np32 = lib.wmh_conv3d_k3_num_partials(cfg0, N, S, S, S)
    part32 = torch.empty(N, Cout, np32, 4, device='cuda')

    def f32():
        _lib.check(lib.wmh_conv3d_k3_fwd(cfg0, P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wf32d), P(bias), P(y32), P(part32), None, N, S, S, S, Cout, 1, 0.01, None), 'f32')
    x3(); f32(); torch.cuda.synchronize()
    msg = '%dx%d^3 %d+%d->%d cfg %d:' % (N, S, C0, C1, Cout, cfg)
    if check:
        act = torch.nn.functional.leaky_relu(x0.double() * ss0[..., 0, None, None, None].double() + ss0[..., 1, None, None, None].double(), 0.01)
        if C1:
            act = torch.cat([act, torch.nn.functional.leaky_relu(x1.double() * ss1[..., 0, None, None, None].double() + ss1[..., 1, None, None, None].double(), 0.01)], 1)
        ref = torch.nn.functional.conv3d(act, torch.from_numpy(w).cuda().double().view(Cout, Cin, 3, 3, 3), bias.double(), padding=1)
        sc = float(ref.abs().max())
        e3, e32 = float((y.double() - ref).abs().max()) / sc, float((y32.double() - ref).abs().max()) / sc
        r3, r32 = float((y.double() - ref).norm() / ref.norm()), float((y32.double() - ref).norm() / ref.norm())
        msg += '  max err / max|ref|: x3 %.3g fp32 %.3g;  rel L2: x3 %.3g fp32 %.3g' % (e3, e32, r3, r32)
    for name, fn in (('x3', x3), ('fp32', f32)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        msg += '  %s %.3f ms' % (name, e0.elapsed_time(e1) / reps)
    print(msg, flush=True)
    if os.environ.get('WMH_STAMPS'):
        stamps = torch.zeros((1 << 20) * 4, dtype=torch.int64, device='cuda')
        lib.wmh_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); x3(); e1.record(); torch.cuda.synchronize()
        lib.wmh_debug_set_conv_stamps(None)
        raw = stamps.view(-1, 8).cpu().numpy()
        st = raw[raw[:, 3] > 0]
        n = st[:, 3].mean()
        nb = len(st)
        print('   loader: issue of loads %.0f, conversion + LDS writes %.0f ticks per iteration' % (raw[nb:2 * nb, 0].mean() / n, raw[nb:2 * nb, 1].mean() / n))
        print('   x3 per chunk iteration of 8 channels (ticks): consumer mfma %.0f + barrier wait %.0f, epilogue %.0f per iteration; loader work %.0f + '
              'barrier wait %.0f; %d workgroups x %.0f iterations' % (st[:, 0].mean() / n, st[:, 1].mean() / n, st[:, 2].mean() / n, st[:, 4].mean() / n,
                                                                     st[:, 5].mean() / n, len(st), n))
        # s_memtime counts shader-clock cycles: a consumer wave's phases add up to the workgroup's life time
        cyc = float((st[:, 0] + st[:, 1] + st[:, 2]).mean())
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        issue = (82.0 / 27.0) * 2 * 27 * Cin * Cout * N * S ** 3 / 32768.0 / (4 * ncu) * 32   # 82 MFMAs per 27 taps and 16 channels; per SIMD
        print('   stamped launch %.3f ms, %.0f cycles per workgroup: the CUs ran at %.0f MHz; MFMA issue alone is %.0f cycles = %.2f of them'
              % (e0.elapsed_time(e1), cyc, cyc / e0.elapsed_time(e1) / 1e3, issue, issue / cyc))


def run_s2(N, S, Cin, Cout, reps=5):
    """The strided layers with 32-voxel output rows (configuration 1, conv3d_k3s2_f16x2_ws.hip): time, and error against float64 at N <= 2."""
    g = torch.Generator(device='cuda').manual_seed(1)
    x0 = torch.randn(N, Cin, S, S, S, device='cuda', generator=g)
    ss0 = torch.stack([1 + 0.3 * torch.randn(N, Cin, device='cuda', generator=g), 0.2 * torch.randn(N, Cin, device='cuda', generator=g)], -1).contiguous()
    w = (torch.randn(Cout, Cin, 27, device='cuda', generator=g) / (27 * Cin) ** 0.5).contiguous()
    bias = torch.randn(Cout, device='cuda', generator=g) * 0.1
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, S, S, S, 2), Cout, S // 2)
    assert cfg == 1 and lib.wmh_conv3d_k3_f16x2_supported(cfg), cfg
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None), 'pack')
    So = S // 2
    y = torch.empty(N, Cout, So, So, So, device='cuda')
    part = torch.empty(N, Cout, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, So, So, So), 4, device='cuda')

    def x3():
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x0), P(ss0), Cin, None, None, 0, P(wf), P(bias), P(y), P(part), None, N, S, S, S, Cout, 0.01, None), 'x3 s2')
    x3(); torch.cuda.synchronize()
    msg = '%dx%d^3 %d->%d stride 2 cfg %d:' % (N, S, Cin, Cout, cfg)
    if N <= 2:
        act = torch.nn.functional.leaky_relu(x0.double() * ss0[..., 0, None, None, None].double() + ss0[..., 1, None, None, None].double(), 0.01)
        ref = torch.nn.functional.conv3d(act, w.double().view(Cout, Cin, 3, 3, 3), bias.double(), padding=1, stride=2)
        msg += '  rel L2 vs float64 %.3g' % float((y.double() - ref).norm() / ref.norm())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x3()
    e1.record(); torch.cuda.synchronize()
    print(msg + '  x3 %.3f ms' % (e0.elapsed_time(e1) / reps), flush=True)
    if os.environ.get('WMH_STAMPS'):
        stamps = torch.zeros((1 << 16) * 8, dtype=torch.int64, device='cuda')
        lib.wmh_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
        e0.record(); x3(); e1.record(); torch.cuda.synchronize()
        lib.wmh_debug_set_conv_stamps(None)
        raw = stamps.view(-1, 8).cpu().numpy().astype('float64')
        st = raw[:256][raw[:256, 3] > 0]
        n = st[:, 3].mean()
        nb = len(st)
        if raw[nb:13 * nb, 6].sum() > 0:
            print('   work per chunk of loader waves 4..11: %s; MFMA phase of consumer waves 0..3: %s, their epilogue per tile: %s'
                  % (' '.join('%.0f' % (raw[(w - 3) * nb:(w - 2) * nb, 6].mean() / n) for w in range(4, 12)),
                     ' '.join('%.0f' % (raw[(9 + w) * nb:(10 + w) * nb, 6].mean() / n) for w in range(4)),
                     ' '.join('%.0f' % (raw[(9 + w) * nb:(10 + w) * nb, 7].mean() / n * (Cin / 8.0)) for w in range(4))))
        cyc = float((st[:, 0] + st[:, 1] + st[:, 2]).mean())
        print('   per chunk iteration (cycles): consumer mfma %.0f + barrier wait %.0f, epilogue %.0f per iteration (%.0f per tile); loader work %.0f + barrier wait %.0f; '
              '%d workgroups x %.0f iterations; the CUs ran at %.0f MHz; 82 MFMAs = 2624 cycles of issue = %.2f of the cycles'
              % (st[:, 0].mean() / n, st[:, 1].mean() / n, st[:, 2].mean() / n, st[:, 2].mean() / n * (Cin / 8.0), st[:, 4].mean() / n, st[:, 5].mean() / n, len(st), n,
                 cyc / e0.elapsed_time(e1) / 1e3, 2624.0 * n / cyc), flush=True)


run_s2(2, 64, 32, 64)
run_s2(8, 128, 32, 64)
run_s2(8, 64, 64, 128)
run(2, 64, 32, 0, 32)
run(1, 64, 32, 32, 32)
run(2, 32, 128, 0, 128)
run(8, 16, 256, 0, 256)
run(8, 128, 32, 0, 32, check=False)
run(8, 128, 32, 32, 32, check=False)
run(8, 64, 64, 0, 64, check=False)
run(8, 64, 64, 64, 64, check=False)
run(8, 32, 128, 0, 128, check=False)
run(8, 32, 128, 128, 128, check=False)
