"""The wave-specialised pair kernel with and without split-K slab memory on the layers whose tiles do not fill the chip (16^3 level at
the training batch, 8^3 ...).  python3 tools/ws_splitk_bench.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
lib = _lib.load(); P = _lib.ptr
LAYERS = [(2, 16, 256, 256, 1), (2, 16, 512, 256, 1), (2, 32, 128, 128, 1), (16, 16, 256, 256, 1), (1, 16, 256, 256, 1), (2, 64, 64, 64, 1),
          (16, 8, 320, 320, 1), (16, 8, 640, 320, 1), (16, 32, 128, 256, 2), (16, 16, 256, 320, 2)]
if os.environ.get('WMH_SMALL_ONLY'):
    LAYERS = LAYERS[6:]
for (N, E, Cin, Cout, stride) in LAYERS:
    D = H = W = E
    cfg = lib.wmh_conv3d_k3_f16x2_config(_lib.check(lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, stride), 'config'), Cout, W)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, D, H, W, generator=g).cuda()
    ss = torch.stack([1 + 0.3 * torch.randn(N, Cin, generator=g), 0.2 * torch.randn(N, Cin, generator=g)], -1).contiguous().cuda()
    w = (torch.randn(Cout, Cin, 27, generator=g) / np.sqrt(27 * Cin)).numpy()
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), np.uint16)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    b = torch.zeros(Cout, device='cuda')
    Do = (D - 1) // stride + 1
    y = torch.empty(N, Cout, Do, Do, Do, device='cuda')
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, N, Do, Do, Do)
    part = torch.empty(N, Cout, npart, 4, device='cuda')
    nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, Cin, N, D, H, W, Cout)
    ws = torch.empty(max(nws, 1), device='cuda')
    res = []
    for use in (False, True):
        def run():
            _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x), P(ss), Cin, None, None, 0, P(wfd), P(b), P(y), P(part), P(ws) if (use and nws) else None,
                                                    N, D, H, W, Cout, 0.01, None), 'fwd')
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    flop = 2.0 * 27 * Cin * Cout * N * Do ** 3
    print('%d x %d^3 (stride %d), %d -> %d (cfg %d): slabs for %d slice(s); without %.3f ms (%.0f TFLOP/s), with %.3f ms (%.0f TFLOP/s)'
          % (N, E, stride, Cin, Cout, cfg, nws // max(y.numel(), 1), res[0], flop / res[0] / 1e9, res[1], flop / res[1] / 1e9))
