"""Cycle stamps of the composed decoder kernel (csrc/conv3d_up_f16x2.hip; diagnostic library): consumer MFMA phase / barrier wait / epilogue and
loader work / barrier wait per chunk iteration, at the three BASELINE stages it serves.  python tools/up_stamps.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib  # noqa: E402
lib = _lib.load_diag()
P = _lib.ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device='cuda').manual_seed(0)
for (Cl, Cu, Cout, S) in ((64, 32, 32, 128), (128, 64, 64, 64), (256, 128, 128, 32)):
    Sl = S // 2
    low = torch.randn(N, Cl, Sl, Sl, Sl, device='cuda', generator=g)
    ss = torch.stack([1 + 0.3 * torch.randn(N, Cl, device='cuda', generator=g), 0.2 * torch.randn(N, Cl, device='cuda', generator=g)], -1).contiguous()
    rng = np.random.default_rng(0)
    w3 = (rng.standard_normal((Cout, 2 * Cu, 27)) / np.sqrt(27 * 2 * Cu)).astype(np.float32)
    wt = (rng.standard_normal((Cl, Cu, 8)) / np.sqrt(Cl)).astype(np.float32)
    wimg = np.zeros(lib.wmh_conv3d_up_packed_bytes_f16x2(Cl, Cout), np.uint8)
    _lib.check(lib.wmh_conv3d_up_pack_weights_f16x2(w3.ctypes.data_as(C.c_void_p), 2 * Cu, Cu, wt.ctypes.data_as(C.c_void_p), Cl, Cout, wimg.ctypes.data_as(C.c_void_p)), 'pack')
    wd = torch.from_numpy(wimg).cuda()
    y = torch.randn(N, Cout, S, S, S, device='cuda', generator=g)
    part = torch.empty(N, Cout, lib.wmh_conv3d_up_num_partials_f16x2(S, S, S), 4, device='cuda')
    for acc in (1, 0):
        def run():
            _lib.check(lib.wmh_conv3d_up_fwd_f16x2(P(low), P(ss), Cl, P(wd), P(y), P(part), N, S, S, S, Cout, 0.01, acc, None), 'up')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device='cuda')
        lib.wmh_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        lib.wmh_debug_set_conv_stamps(None)
        raw = stamps.view(-1, 8).cpu().numpy().astype('float64')
        st = raw[raw[:, 3] > 0]
        nb = len(st)
        ne = st[:, 3].mean()
        print('   loader, per iteration: load_ss + issue %.0f, convert %.0f, weight write %.0f cycles'
              % (raw[nb:2 * nb, 0].mean() / ne, raw[nb:2 * nb, 1].mean() / ne, raw[nb:2 * nb, 2].mean() / ne))
        n = st[:, 3].mean()
        cyc = (st[:, 0] + st[:, 1] + st[:, 2]).mean()
        nch = (Cl + 7) // 8
        print('%d x %d^3, low %d ch -> %d, accumulate %d: %.3f ms (stamped launch %.3f ms, %.0f MHz); per chunk iteration: consumer mfma %.0f + barrier wait %.0f, '
              'epilogue %.0f per tile (%d chunks); loader work %.0f + barrier wait %.0f; MFMA issue alone %d per chunk'
              % (N, S, Cl, Cout, acc, ms, e0.elapsed_time(e1), cyc / e0.elapsed_time(e1) / 1e3, st[:, 0].mean() / n, st[:, 1].mean() / n, st[:, 2].mean() / n * nch, nch,
                 st[:, 4].mean() / n, st[:, 5].mean() / n, 96 * 32), flush=True)
    del low, y
    torch.cuda.empty_cache()
