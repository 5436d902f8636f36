import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepwmh_amd import _lib
lib = _lib.load_diag() if os.environ.get('WMH_DIAG_LIB') else _lib.load()
n, C, P = 8, 32, (128, 128, 128)
x = torch.randn(n, C, *P, device='cuda'); ss = torch.rand(n, C, 2, device='cuda') + 0.5
w = torch.randn(2, C, device='cuda'); g = torch.rand(*P, device='cuda'); agg = torch.zeros(2, 160, 192, 160, device='cuda')
for flips in ([0] * 8, list(range(8))):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.wmh_seghead_softmax_accumulate(_lib.ptr(x), _lib.ptr(ss), _lib.ptr(w), 2, _lib.ptr(g), _lib.ptr(agg), C, *P, 160, 192, 160,
                                                      _lib.iarr((0, 64, 32)), n, _lib.iarr(flips), 0.01, None), 'sh')
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    gb = n * C * 128 ** 3 * 4 / 1e9
    print('seghead flips %s: %.3f ms  %.2f TB/s' % (flips[:3], np.median(ts), gb / np.median(ts)))
