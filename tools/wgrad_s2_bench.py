"""The weight gradient of the strided layers (batch 2, BASELINE plan): the pair kernel (wmh_conv3d_k3s2_wgrad_f16x2) beside the exact-fp32 kernel that
served them until round 5.  python3 tools/wgrad_s2_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
lib = _lib.load(); P = _lib.ptr
N = 2
for (Cin, Cout, E) in [(32, 64, 128), (64, 128, 64), (128, 256, 32)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, Cin, E, E, E, device='cuda', generator=g)
    ss = torch.stack([1 + 0.1 * torch.randn(N, Cin, device='cuda', generator=g), 0.1 * torch.randn(N, Cin, device='cuda', generator=g)], -1).contiguous()
    dy = torch.randn(N, Cout, E // 2, E // 2, E // 2, device='cuda', generator=g)
    dW = torch.empty(Cout, Cin, 27, device='cuda')
    ws = torch.empty(lib.wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats(Cin, Cout, N, E, E, E), device='cuda')
    ws32 = torch.empty(lib.wmh_conv3d_k3_wgrad_workspace_floats(Cin, Cout, N, E, E, E, 2), device='cuda')
    def pair(): _lib.check(lib.wmh_conv3d_k3s2_wgrad_f16x2(P(x), P(ss), Cin, None, None, 0, P(dy), P(dW), P(ws), N, E, E, E, Cout, 0.01, None), 'pair')
    def fp32(): _lib.check(lib.wmh_conv3d_k3_wgrad(P(x), P(ss), Cin, None, None, 0, P(dy), P(dW), P(ws32), N, E, E, E, Cout, 2, 0.01, None), 'fp32')
    res = []
    for f in (pair, fp32):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    flop = 2.0 * 27 * Cin * Cout * N * (E // 2) ** 3
    print('%d -> %d from 2 x %d^3: pair %.3f ms (%.0f TFLOP/s), exact fp32 %.3f ms (%.0f TFLOP/s)' % (Cin, Cout, E, res[0], flop / res[0] / 1e9, res[1], flop / res[1] / 1e9))
