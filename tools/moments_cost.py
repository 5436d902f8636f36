"""What the InstanceNorm moments in the pair kernels' epilogue cost: the same launch with and without moment records (partials = NULL), at the three
large levels.  Round 5: 1.4-2.6 % of a launch (5.29 / 5.22 ms at 16 x 128^3) -- the epilogue's time is its stores, not the reduction.
    python tools/moments_cost.py"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
lib = _lib.load(); P = _lib.ptr
for (N, S, Cin, Cout) in ((16, 128, 32, 32), (16, 64, 64, 64), (16, 32, 128, 128)):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, Cin, S, S, S, device='cuda', generator=g)
    ss = torch.stack([1 + 0.3 * torch.randn(N, Cin, device='cuda', generator=g), 0.2 * torch.randn(N, Cin, device='cuda', generator=g)], -1).contiguous()
    w = (torch.randn(Cout, Cin, 27, device='cuda', generator=g) / (27 * Cin) ** 0.5).contiguous()
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, S, S, S, 1), Cout, S)
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None), 'pack')
    y = torch.empty(N, Cout, S, S, S, device='cuda')
    part = torch.empty(N, Cout, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, S, S, S), 4, device='cuda')
    res = []
    for pp in (part, None, part, None):
        def run():
            _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x), P(ss), Cin, None, None, 0, P(wf), None, P(y), P(pp), None, N, S, S, S, Cout, 0.01, None), 'x3')
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print('%d x %d^3 %d->%d cfg %d: with moments %.3f / %.3f ms, without %.3f / %.3f ms' % (N, S, Cin, Cout, cfg, res[0], res[2], res[1], res[3]), flush=True)
    del x, y
