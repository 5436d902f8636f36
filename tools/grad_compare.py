"""Gradients of one training iteration at full size (BASELINE plan, batch 2 x 128^3) in the f16x2 and bf16 modes against the fp32 mode:
relative L2 difference over the whole arena and the worst parameter.  python tools/grad_compare.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.synthetic import he_init_state_dict
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
plan = baseline_plan()
sd = he_init_state_dict(plan, 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 1, 128, 128, 128, generator=g).cuda()
lab = (torch.rand(2, 128, 128, 128, generator=g) < 0.05).to(torch.uint8).cuda()
td = downsample_targets(lab, plan.num_pool)
res = {}
for prec in ('fp32', 'f16x2'):
    tr = UNetTrainer(plan, sd, 'cuda:0', precision=prec)
    tr.forward(x); l = float(tr.backward(td).cpu())
    res[prec] = (l, {k: v.clone() for k, v in tr.g.items()})
    del tr; torch.cuda.empty_cache()
for prec in ('f16x2',):
    num = den = 0.0; worst = ('', 0.0)
    for k, v in res['fp32'][1].items():
        a = res[prec][1][k]
        n2, d2 = float(((a - v) ** 2).sum()), float((v ** 2).sum())
        num += n2; den += d2
        r = (n2 / max(d2, 1e-60)) ** 0.5
        if r > worst[1]: worst = (k, r)
    print('%s: loss %.7f (fp32 %.7f)  gradient rel L2 vs fp32 %.3g  worst parameter %s %.3g' % (prec, res[prec][0], res['fp32'][0], (num / den) ** 0.5, worst[0], worst[1]))
