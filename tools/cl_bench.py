"""Per-layer timing of the channels-last bf16 kernels at the BASELINE shapes (batch 8), HIP events, inputs resident.
python tools/cl_bench.py [reps]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 8


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rnd_cl(n, d, h, w, c):
    return torch.randn(n, d, h, w, c, device='cuda').to(torch.bfloat16)


def conv_layer(name, S, C0, C1, Cout, stride=1):
    Cin = C0 + C1
    x0 = rnd_cl(N, S, S, S, C0)
    x1 = rnd_cl(N, S, S, S, C1) if C1 else None
    ss0 = torch.randn(N, C0, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')
    ss1 = (torch.randn(N, C1, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')) if C1 else None
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 27)) / np.sqrt(27 * Cin)).astype(np.float32)
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(0, Cin, Cout), dtype=np.uint16)
    lib.wmh_conv3d_k3_pack_weights_bf16(0, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    bias = torch.zeros(Cout, device='cuda')
    So = S // stride
    y = torch.empty(N, So, So, So, Cout, dtype=torch.bfloat16, device='cuda')
    if stride == 1:
        npart = lib.wmh_cl_conv3_num_partials(S, S, S)
        part = torch.empty(N, Cout, npart, 4, device='cuda')
        fn = lambda: _lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wfd), _lib.ptr(bias),
                                                     _lib.ptr(y), _lib.ptr(part), N, S, S, S, Cout, 0.01, None), name)
    else:
        npart = lib.wmh_cl_conv3s2_num_partials(So, So, So)
        part = torch.empty(N, Cout, npart, 4, device='cuda')
        fn = lambda: _lib.check(lib.wmh_cl_conv3s2_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(wfd), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(part),
                                                       N, S, S, S, Cout, 0.01, None), name)
    ms = timeit(fn)
    flop = 2.0 * Cin * Cout * 27 * N * So ** 3
    nbytes = 2.0 * N * Cin * S ** 3 + 2.0 * N * Cout * So ** 3
    print('%-28s %7.3f ms  %7.1f TFLOP/s (%.3f of 2500)  %6.2f TB/s algorithmic (%.3f of 8)'
          % (name, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8), flush=True)
    return ms


def first_layer():
    S = 128
    x = torch.randn(N, 1, S, S, S, device='cuda')
    w = np.random.default_rng(0).standard_normal((32, 1, 27)).astype(np.float32)
    wf = np.empty(lib.wmh_cl_conv3_first_packed_halfs(32), dtype=np.uint16)
    lib.wmh_cl_conv3_first_pack_weights(w.ctypes.data_as(C.c_void_p), 32, wf.ctypes.data_as(C.c_void_p))
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    npart = lib.wmh_cl_conv3_first_num_partials(S, S, S)
    y = torch.empty(N, S, S, S, 32, dtype=torch.bfloat16, device='cuda')
    part = torch.empty(N, 32, npart, 4, device='cuda')
    ms = timeit(lambda: _lib.check(lib.wmh_cl_conv3_first_fwd(_lib.ptr(x), _lib.ptr(wfd), None, _lib.ptr(y), _lib.ptr(part), N, S, S, S, 32, None), 'first'))
    nbytes = 4.0 * x.numel() + 2.0 * y.numel()
    print('%-28s %7.3f ms  %6.2f TB/s algorithmic (%.3f of 8)' % ('enc0.0 1->32 @128', ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8), flush=True)
    return ms


def convT(name, S, Cin, Cout):
    x = rnd_cl(N, S, S, S, Cin)
    ss = torch.randn(N, Cin, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')
    w = np.random.default_rng(0).standard_normal((Cin, Cout, 8)).astype(np.float32)
    wf = np.empty(lib.wmh_cl_convt_packed_halfs(Cin, Cout), dtype=np.uint16)
    lib.wmh_cl_convt_pack_weights(w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    y = torch.empty(N, 2 * S, 2 * S, 2 * S, Cout, dtype=torch.bfloat16, device='cuda')
    ms = timeit(lambda: _lib.check(lib.wmh_cl_convt_fwd(_lib.ptr(x), _lib.ptr(ss), _lib.ptr(wfd), _lib.ptr(y), N, Cin, Cout, S, S, S, 0.01, None), name))
    nbytes = 2.0 * x.numel() + 2.0 * y.numel()
    print('%-28s %7.3f ms  %6.2f TB/s algorithmic (%.3f of 8)' % (name, ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8), flush=True)
    return ms


def head():
    P = 128
    x = rnd_cl(8, P, P, P, 32)
    ss = torch.randn(8, 32, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')
    wseg = torch.randn(2, 32, device='cuda')
    gauss = torch.rand(P, P, P, device='cuda')
    agg = torch.zeros(2, 160, 192, 160, device='cuda')
    ms = timeit(lambda: _lib.check(lib.wmh_cl_seghead_softmax_accumulate(_lib.ptr(x), _lib.ptr(ss), _lib.ptr(wseg), 2, _lib.ptr(gauss), _lib.ptr(agg), 32,
                                                                         P, P, P, 160, 192, 160, _lib.iarr((0, 0, 0)), 8, _lib.iarr(range(8)), 0.01, None), 'head'))
    nbytes = 2.0 * x.numel() + 4.0 * 2 * P ** 3 * 2 + 4.0 * P ** 3
    print('%-28s %7.3f ms  %6.2f TB/s algorithmic (%.3f of 8)' % ('head + softmax + accumulate', ms, nbytes / ms / 1e9, nbytes / ms / 1e9 / 8), flush=True)
    return ms


tot = 0.0
tot += first_layer()
tot += conv_layer('enc0.1 32->32 @128', 128, 32, 0, 32)
tot += conv_layer('enc1.0 32->64 @128 s2', 128, 32, 0, 64, 2)
tot += conv_layer('enc1.1 64->64 @64', 64, 64, 0, 64)
tot += conv_layer('enc2.0 64->128 @64 s2', 64, 64, 0, 128, 2)
tot += conv_layer('enc2.1 128->128 @32', 32, 128, 0, 128)
tot += conv_layer('dec2.0 128+128->128 @32', 32, 128, 128, 128)
tot += conv_layer('dec2.1 128->128 @32', 32, 128, 0, 128)
tot += convT('tu3 128->64 @32->64', 32, 128, 64)
tot += conv_layer('dec3.0 64+64->64 @64', 64, 64, 64, 64)
tot += conv_layer('dec3.1 64->64 @64', 64, 64, 0, 64)
tot += convT('tu4 64->32 @64->128', 64, 64, 32)
tot += conv_layer('dec4.0 32+32->32 @128', 128, 32, 32, 32)
tot += conv_layer('dec4.1 32->32 @128', 128, 32, 0, 32)
tot += head()
print('sum of the channels-last layers: %.2f ms per batch-8 forward' % tot)
