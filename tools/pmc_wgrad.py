"""Workload for the HBM-traffic PMC passes of the pair weight-gradient kernel (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
the calibration launch of tools/pmc_traffic.py (1 GiB in + 1 GiB out, 4 B per lane), then the kernel on three of the training step's
layers at batch 2.  WMH_WGRAD_PLAIN_ORDER=1 selects the plain x -> y -> z tile order.  tools/parse_pmc.py reads the CSVs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from deepwmh_amd import _lib  # noqa: E402

lib = _lib.load_diag() if (os.environ.get('WMH_STAMPS') or os.environ.get('WMH_DIAG_LIB') or os.environ.get('WMH_WGRAD_PLAIN_ORDER')) else _lib.load()   # stamps and variant switches: diagnostic library only
P = _lib.ptr
n = 256 * 1024 * 1024
a = torch.randn(n, device='cuda')
b = torch.empty_like(a)
ss = torch.tensor([[1.5, 0.25]], device='cuda')
for _ in range(2):
    _lib.check(lib.wmh_instnorm_lrelu_apply(P(a), P(ss), P(b), 1, 1, n, 0.01, None), 'apply')
torch.cuda.synchronize()
del a, b
for N, C0, Cout, S in ((2, 32, 32, 128), (2, 64, 32, 128), (2, 128, 64, 64)):
    x = torch.randn(N, C0, S, S, S, device='cuda')
    s0 = torch.stack([0.5 + torch.rand(N, C0, device='cuda'), 0.3 * torch.randn(N, C0, device='cuda')], -1)
    dy = torch.randn(N, Cout, S, S, S, device='cuda')
    dW = torch.empty(Cout, C0, 27, device='cuda')
    ws = torch.empty(lib.wmh_conv3d_k3_wgrad_f16x2_workspace_floats(C0, Cout, N, S, S, S), device='cuda')
    for _ in range(2):
        _lib.check(lib.wmh_conv3d_k3_wgrad_f16x2(P(x), P(s0), C0, None, None, 0, P(dy), P(dW), P(ws), N, S, S, S, Cout, 0.01, None), 'wgrad')
    torch.cuda.synchronize()
    print('%d x %d^3 %d->%d: algorithmic bytes %.3f GB (x + dy read once)' % (N, S, C0, Cout, 4.0 * N * (C0 + Cout) * S ** 3 / 1e9))
print('done')
