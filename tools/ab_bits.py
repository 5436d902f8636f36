"""Bit comparison of two builds of the library on the pair convolution (outputs and InstanceNorm moment records), e.g. the interleaved
epilogue against the plain one:  python tools/ab_bits.py <lib A .so> <lib B .so>"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib  # noqa: E402


def open_lib(path):
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


A, B = open_lib(sys.argv[1]), open_lib(sys.argv[2])
P = _lib.ptr
g = torch.Generator(device='cuda').manual_seed(7)
bad = 0
for (N, C0, C1, Cout, D, H, W) in [(2, 32, 0, 32, 64, 64, 64), (1, 32, 32, 32, 24, 40, 64), (2, 16, 0, 32, 9, 20, 72), (3, 24, 8, 64, 13, 33, 36), (8, 32, 0, 32, 128, 128, 128)]:
    Cin = C0 + C1
    x0 = torch.randn(N, C0, D, H, W, device='cuda', generator=g)
    x1 = torch.randn(N, C1, D, H, W, device='cuda', generator=g) if C1 else None
    s0 = torch.stack([1 + 0.3 * torch.randn(N, C0, device='cuda', generator=g), 0.2 * torch.randn(N, C0, device='cuda', generator=g)], -1).contiguous()
    s1 = torch.stack([1 + 0.3 * torch.randn(N, C1, device='cuda', generator=g), 0.2 * torch.randn(N, C1, device='cuda', generator=g)], -1).contiguous() if C1 else None
    w = (torch.randn(Cout, Cin, 27, device='cuda', generator=g) / (27 * Cin) ** 0.5).contiguous()
    bias = torch.randn(Cout, device='cuda', generator=g) * 0.1
    outs = []
    for lib in (A, B):
        cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, D, H, W, 1), Cout, W)
        assert lib.wmh_conv3d_k3_f16x2_supported(cfg), cfg
        wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), dtype=torch.int16, device='cuda')
        assert lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None) == 0
        y = torch.full((N, Cout, D, H, W), float('nan'), device='cuda')
        npart = lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, D, H, W)
        part = torch.zeros(N, Cout, npart, 4, device='cuda')
        nws = lib.wmh_conv3d_k3_f16x2_workspace_floats(cfg, Cin, N, D, H, W, Cout)
        ws = torch.empty(max(nws, 1), device='cuda')
        rc = lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x0), P(s0), C0, P(x1), P(s1), C1, P(wf), P(bias), P(y), P(part), P(ws) if nws else None, N, D, H, W, Cout, 0.01, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append((cfg, y, part))
    same_y, same_p = torch.equal(outs[0][1], outs[1][1]), torch.equal(outs[0][2], outs[1][2])
    fin = bool(torch.isfinite(outs[0][1]).all())
    print('%dx(%d+%d)->%d %dx%dx%d cfg %d: outputs %s, moment records %s, finite %s' % (N, C0, C1, Cout, D, H, W, outs[0][0], 'equal' if same_y else 'DIFFER',
                                                                                 'equal' if same_p else 'DIFFER', fin), flush=True)
    if not same_p:
        d = (outs[0][2] - outs[1][2]).abs()
        print('   max record difference', float(d.max()), 'of', float(outs[1][2].abs().max()))
    bad += (not same_y) + (not same_p) + (not fin)
sys.exit(1 if bad else 0)
