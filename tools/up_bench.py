"""A decoder stage's first convolution, two ways, at the BASELINE shapes (f16x2): (a) transposed convolution + two-source 3x3x3 pair kernel,
(b) pair kernel on the skip channels + the composed kernel of csrc/conv3d_up_f16x2.hip.  Times per launch and the difference of the results.
    python tools/up_bench.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd.network import UNetEngine  # noqa: E402
from deepwmh_amd.plan import baseline_plan  # noqa: E402
from deepwmh_amd.synthetic import he_init_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
plan = baseline_plan()
sd = he_init_state_dict(plan, 0)
engs = {False: UNetEngine(plan, sd, 'cuda:0', precision='f16x2', fuse_up=False), True: UNetEngine(plan, sd, 'cuda:0', precision='f16x2', fuse_up=True)}
g = torch.Generator(device='cuda').manual_seed(0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


nd = len(engs[True].dec)
for u in range(nd - 1, -1, -1):
    st = engs[True].dec[u]
    S = 128 >> (nd - 1 - u)
    Sl = S // 2
    if Sl < 4:
        continue
    Cl, Cs = st['cdown'], st['cskip']
    low = torch.randn(N, Cl, Sl, Sl, Sl, device='cuda', generator=g)
    low_ss = torch.stack([1 + 0.3 * torch.randn(N, Cl, device='cuda', generator=g), 0.2 * torch.randn(N, Cl, device='cuda', generator=g)], -1).contiguous()
    skip = torch.randn(N, Cs, S, S, S, device='cuda', generator=g)
    skip_ss = torch.stack([1 + 0.3 * torch.randn(N, Cs, device='cuda', generator=g), 0.2 * torch.randn(N, Cs, device='cuda', generator=g)], -1).contiguous()
    res = {}
    for fused, eng in engs.items():
        stg = eng.dec[u]

        def old():
            up = eng._convT('tu%d' % u, stg['tu_w'], stg['tu_wf'], low, low_ss, stg['cskip'], stg['kt'], wf_x3=stg['tu_wf_x3'])
            return eng._conv('dec%d.0' % u, stg['convs'][0], up, None, skip, skip_ss)

        def new():
            return eng._dec_first_fused('dec%d.0' % u, stg, low, low_ss, skip, skip_ss)
        if fused and new() is None:
            print('stage %d (%d^3): composed kernel not available' % (u, S)); continue
        ms, (y, ss) = timed(new if fused else old)
        res[fused] = (ms, y.clone(), ss.clone())
    if len(res) == 2:
        d = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
        dss = float((res[True][2] - res[False][2]).abs().max())
        print('stage %d: %d x %d^3, low %d ch -> up %d + skip %d -> %d: two kernels + up tensor %.3f ms, composed %.3f ms; max |dy| / max |y| %.3g, max |d(scale, shift)| %.3g'
              % (u, N, S, Cl, Cs, Cs, st['convs'][0].cout, res[False][0], res[True][0], d, dss), flush=True)
    del low, skip
    torch.cuda.empty_cache()
