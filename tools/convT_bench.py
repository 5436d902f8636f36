"""Per-layer timing of the pair-mode transposed convolution (wmh_convtranspose3d_fwd_f16x2) at the inference batch (16 patches) against
the bytes it moves (input once, output once).  usage: python tools/convT_bench.py   (WMH_LIB_PATH selects the library build)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib  # noqa: E402

lib = _lib.load()
P = _lib.ptr
N = int(os.environ.get('WMH_BATCH', '16'))
g = torch.Generator(device='cuda').manual_seed(5)
tot = 0.0
for Cin, Cout, E in [(64, 32, 64), (128, 64, 32), (256, 128, 16), (320, 256, 8), (320, 320, 4)]:
    x = torch.randn(N, Cin, E, E, E, device='cuda', generator=g)
    ss = torch.stack([1 + 0.1 * torch.randn(N, Cin, device='cuda', generator=g), 0.1 * torch.randn(N, Cin, device='cuda', generator=g)], -1).contiguous()
    w = (torch.randn(Cin, Cout, 2, 2, 2, device='cuda', generator=g) / (8 * Cin) ** 0.5).contiguous()
    wf = torch.empty(lib.wmh_convtranspose3d_packed_dwords_f16x2(Cin, Cout, 2), dtype=torch.int32, device='cuda')
    _lib.check(lib.wmh_convtranspose3d_pack_weights_f16x2_device(P(w), Cin, Cout, 2, P(wf), None), 'pack')
    y = torch.empty(N, Cout, 2 * E, 2 * E, 2 * E, device='cuda')

    def run():
        _lib.check(lib.wmh_convtranspose3d_fwd_f16x2(P(x), P(ss), P(wf), P(y), N, Cin, Cout, E, E, E, 2, 0.01, None), 'convT')
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = (x.numel() + y.numel()) * 4 / 1e9
    tot += ms
    print('%dx%d->%d @%d^3: %.3f ms  %.2f GB -> %.2f TB/s   checksum %.6e' % (N, Cin, Cout, E, ms, gb, gb / ms, float(y.double().sum())), flush=True)
print('sum: %.3f ms' % tot)
