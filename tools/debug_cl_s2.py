"""Diagnostic: stride-2 CL16 conv with delta weights, to see which input voxel / channel reaches which output."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
lib = _lib.load()
N, D, H, W, Cin, Cout = 1, 4, 4, 64, 32, 32
rng = np.random.default_rng(0)
x = rng.integers(-3, 4, (N, Cin, D, H, W)).astype(np.float32)
xc = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 4, 1))).to(torch.bfloat16).cuda()
for tap in (13, 12, 14, 10, 16, 4, 22, 0, 26):
    w = np.zeros((Cout, Cin, 27), np.float32)
    for c in range(32):
        w[c, c, tap] = 1.0
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(0, Cin, Cout), dtype=np.uint16)
    lib.wmh_conv3d_k3_pack_weights_bf16(0, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    Do, Ho, Wo = 2, 2, 32
    y = torch.full((N, Do, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.wmh_cl_conv3s2_fwd(_lib.ptr(xc), None, Cin, _lib.ptr(wfd), None, _lib.ptr(y), None, N, D, H, W, Cout, 0.01, None), 's2')
    got = y.float().cpu().numpy().transpose(0, 4, 1, 2, 3)
    kd, kr, kw = tap // 9, (tap // 3) % 3, tap % 3
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1), (1, 1)))
    ref = xp[:, :, kd:kd + 2 * Do:2, kr:kr + 2 * Ho:2, kw:kw + 2 * Wo:2]
    bad = got != ref
    print('tap %2d (kd %d kr %d kw %d): mismatches %d of %d' % (tap, kd, kr, kw, int(bad.sum()), bad.size))
    if bad.any() and tap == 13:
        idx = np.argwhere(bad)[:6]
        for i in idx:
            print('   at', tuple(i), 'got', got[tuple(i)], 'ref', ref[tuple(i)])
        # where does got[c, 0,0,:] come from?
        print('   got[0,0,0,0,:8]', got[0, 0, 0, 0, :8], 'ref', ref[0, 0, 0, 0, :8])
        print('   got[0,5,1,1,:8]', got[0, 5, 1, 1, :8], 'ref', ref[0, 5, 1, 1, :8])
