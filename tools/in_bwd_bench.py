"""Per-layer timing of the InstanceNorm + LeakyReLU backward (wmh_in_lrelu_bwd: reduce, combine, apply, parameter gradients) on the
BASELINE plan's training shapes (batch 2), against the bytes it has to move: y and the upstream gradient(s) twice, dy once.
usage: python tools/in_bwd_bench.py            (WMH_LIB_PATH selects the library build)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib  # noqa: E402

lib = _lib.load()
P = _lib.ptr
N = 2
# (channels, edge, number of upstream gradient tensors, layers of that kind in the network[, head folded in: the stage's deep-supervision logits])
LAYERS = [(32, 128, 1, 2), (32, 128, 2, 1), (64, 64, 1, 2), (64, 64, 2, 1), (128, 32, 1, 2), (128, 32, 2, 1), (256, 16, 1, 2), (256, 16, 2, 1),
          (320, 8, 1, 3), (320, 8, 2, 1), (320, 4, 1, 2),
          (32, 128, 1, 1, True), (64, 64, 1, 1, True), (128, 32, 1, 1, True), (256, 16, 1, 1, True)]
g = torch.Generator(device='cuda').manual_seed(3)
total = 0.0
floor = 0.0
for C, E, ng, count, *head in LAYERS:
    head = bool(head and head[0])
    S = E ** 3
    y = torch.randn(N, C, S, device='cuda', generator=g)
    g1 = torch.randn(N, C, S, device='cuda', generator=g)
    g2 = torch.randn(N, C, S, device='cuda', generator=g) if ng == 2 else None
    ss = torch.stack([1 + 0.1 * torch.randn(N, C, device='cuda', generator=g), 0.1 * torch.randn(N, C, device='cuda', generator=g)], -1).contiguous()
    mr = torch.stack([0.1 * torch.randn(N, C, device='cuda', generator=g), 1 + 0.1 * torch.rand(N, C, device='cuda', generator=g)], -1).contiguous()
    gamma = torch.ones(C, device='cuda')
    dy = torch.empty_like(y)
    dg, db, dbias = (torch.empty(C, device='cuda') for _ in range(3))
    dl = torch.randn(N, 2, S, device='cuda', generator=g) if head else None
    wseg = torch.randn(2, C, device='cuda', generator=g) if head else None
    dwseg = torch.empty(2, C, device='cuda') if head else None
    scratch = torch.empty(lib.wmh_in_lrelu_bwd_scratch_doubles(N, C, S), dtype=torch.float64, device='cuda')

    def run():
        _lib.check(lib.wmh_in_lrelu_bwd(P(y), P(ss), P(mr), P(gamma), P(g1), P(g2) if g2 is not None else None, P(dl) if head else None, P(wseg) if head else None,
                                        2 if head else 0, P(dy), P(dg), P(db), P(dwseg) if head else None, P(dbias), P(scratch), N, C, S, 0.01, None), 'in_lrelu_bwd')
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = N * C * S * 4 * (2 * (1 + ng) + 1) / 1e9
    total += ms * count
    floor += gb / 6.3 * count
    print('2x%dx%d^3, %d upstream%s: %.3f ms  %.2f GB -> %.2f TB/s  (x%d layers)' % (C, E, ng, ' + head' if head else '', ms, gb, gb / ms, count), flush=True)
print('sum over the network: %.2f ms; the same bytes at 6.3 TB/s: %.2f ms' % (total, floor))
