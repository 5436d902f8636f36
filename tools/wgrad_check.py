"""Weight-gradient kernels on the BASELINE plan's stride-1 layers (batch 2 x 128^3 training patch): time per launch of the exact-fp32
kernel, the bf16-operand kernel and the f16x2 pair kernel, error of each against float64 on a small layer, and -- WMH_STAMPS=1 -- the
pair kernel's consumer / loader cycle split.

    python tools/wgrad_check.py            (GPU)
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from deepwmh_amd import _lib  # noqa: E402

SLOPE = 0.01
P = _lib.ptr


def run(lib, mode, d, C0, C1, Cout, N, D, H, W, dW, ws, stamps=None):
    Cin = C0 + C1
    if mode == 'fp32':
        _lib.check(lib.wmh_conv3d_k3_wgrad(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, 1, SLOPE, None), mode)
    elif mode == 'bf16':
        _lib.check(lib.wmh_conv3d_k3_wgrad_bf16(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, SLOPE, None), mode)
    elif stamps is not None:
        _lib.check(lib.wmh_conv3d_k3_wgrad_f16x2_stamped(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, SLOPE,
                                                         P(stamps), None), mode)
    else:
        _lib.check(lib.wmh_conv3d_k3_wgrad_f16x2(P(d[0]), P(d[1]), C0, P(d[2]), P(d[3]), C1, P(d[4]), P(dW), P(ws), N, D, H, W, Cout, SLOPE, None), mode)


def workspace(lib, mode, Cin, Cout, N, D, H, W):
    if mode == 'fp32':
        return lib.wmh_conv3d_k3_wgrad_workspace_floats(Cin, Cout, N, D, H, W, 1)
    if mode == 'bf16':
        return lib.wmh_conv3d_k3_wgrad_bf16_workspace_floats(Cin, Cout, N, D, H, W)
    return lib.wmh_conv3d_k3_wgrad_f16x2_workspace_floats(Cin, Cout, N, D, H, W)


def main():
    lib = _lib.load_diag() if (os.environ.get('WMH_STAMPS') or os.environ.get('WMH_DIAG_LIB')) else _lib.load()   # stamps: diagnostic library only
    dev = 'cuda'
    g = torch.Generator().manual_seed(5)
    # accuracy on a layer float64 autograd finishes quickly
    N, C0, C1, Cout, D, H, W = 2, 32, 0, 32, 16, 32, 64
    x0 = torch.randn(N, C0, D, H, W, generator=g); s0 = torch.stack([0.5 + torch.rand(N, C0, generator=g), 0.3 * torch.randn(N, C0, generator=g)], -1)
    dy = torch.randn(N, Cout, D, H, W, generator=g)
    act = F.leaky_relu(x0 * s0[..., 0][:, :, None, None, None] + s0[..., 1][:, :, None, None, None], SLOPE)
    w = torch.zeros(Cout, C0, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(act.double(), w, None, padding=1).backward(dy.double())
    ref = w.grad.reshape(Cout, C0, 27)
    d = [x0.to(dev), s0.to(dev), None, None, dy.to(dev)]
    line = '%dx%dx%dx%d %d->%d vs float64:' % (N, D, H, W, C0, Cout)
    for mode in ('fp32', 'bf16', 'f16x2'):
        dW = torch.empty(Cout, C0, 27, device=dev)
        ws = torch.empty(workspace(lib, mode, C0, Cout, N, D, H, W), device=dev)
        run(lib, mode, d, C0, 0, Cout, N, D, H, W, dW, ws)
        torch.cuda.synchronize()
        e = (dW.cpu().double() - ref)
        line += '  %s rel. L2 %.2e max/max %.2e' % (mode, float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()))
    print(line)
    # speed on the training step's layers
    layers = [(2, 1, 0, 32, 128), (2, 32, 0, 32, 128), (2, 32, 32, 32, 128), (2, 32, 0, 64, 64), (2, 64, 0, 64, 64), (2, 64, 64, 64, 64),
              (2, 128, 0, 128, 32), (2, 128, 128, 128, 32), (2, 256, 0, 256, 16), (2, 256, 256, 256, 16)]
    if len(sys.argv) > 1:
        layers = [layers[int(v)] for v in sys.argv[1:]]
    stamping = bool(os.environ.get('WMH_STAMPS'))
    tot = {'fp32': 0.0, 'bf16': 0.0, 'f16x2': 0.0}
    for N, C0, C1, Cout, S in layers:
        D = H = W = S
        Cin = C0 + C1
        x0 = torch.randn(N, C0, D, H, W, device=dev); s0 = torch.stack([0.5 + torch.rand(N, C0, device=dev), 0.3 * torch.randn(N, C0, device=dev)], -1)
        x1 = torch.randn(N, C1, D, H, W, device=dev) if C1 else None
        s1 = torch.stack([0.5 + torch.rand(N, C1, device=dev), 0.3 * torch.randn(N, C1, device=dev)], -1) if C1 else None
        dy = torch.randn(N, Cout, D, H, W, device=dev)
        d = [x0, s0, x1, s1, dy]
        flops = 2.0 * 27 * Cin * Cout * N * D * H * W
        line = '%dx%d^3 %d+%d->%d:' % (N, S, C0, C1, Cout)
        outs = {}
        for mode in ('fp32', 'bf16', 'f16x2'):
            dW = torch.empty(Cout, Cin, 27, device=dev)
            ws = torch.empty(workspace(lib, mode, Cin, Cout, N, D, H, W), device=dev)
            for _ in range(2):
                run(lib, mode, d, C0, C1, Cout, N, D, H, W, dW, ws)
            torch.cuda.synchronize()
            reps = int(os.environ.get('WMH_REPS', '10'))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(lib, mode, d, C0, C1, Cout, N, D, H, W, dW, ws)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot[mode] += ms
            outs[mode] = dW.clone()
            line += '  %s %.3f ms (%.0f TFLOP/s)' % (mode, ms, flops / ms / 1e9)
        r = float((outs['f16x2'] - outs['fp32']).norm() / outs['fp32'].norm())
        print(line + '  f16x2 vs fp32 rel. L2 %.1e' % r)
        if stamping:
            n = lib.wmh_conv3d_k3_wgrad_f16x2_stamp_count(Cin, Cout, N, D, H, W)
            stamps = torch.zeros(n, dtype=torch.int64, device=dev)
            dW = torch.empty(Cout, Cin, 27, device=dev)
            ws = torch.empty(workspace(lib, 'f16x2', Cin, Cout, N, D, H, W), device=dev)
            run(lib, 'f16x2', d, C0, C1, Cout, N, D, H, W, dW, ws, stamps)
            torch.cuda.synchronize()
            st = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
            ntiles = N * -(-W // 32 if W >= 24 else -(-W // 16)) * -(-H // (4 if W >= 24 else 8)) * -(-D // 2)
            units = 2.0 * ntiles * (-(-Cout // 32)) * (-(-Cin // 16)) / st.shape[0] if False else None
            print('   pair kernel, ticks per workgroup (mean over %d): consumer MFMA phases %.0f + barrier wait %.0f; loader work %.0f + barrier wait %.0f'
                  % (st.shape[0], st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean(), st[:, 3].mean()))
    print('sum over the listed layers: fp32 %.2f ms, bf16 %.2f ms, f16x2 %.2f ms' % (tot['fp32'], tot['bf16'], tot['f16x2']))


if __name__ == '__main__':
    main()
