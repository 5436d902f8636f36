"""Diagnostic: full-width gradients of the HIP trainer and of torch-CPU fp32 autograd, both against float64 autograd."""
import os, sys, copy, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd.plan import UNetPlan
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
from oracle import training as ot
from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
from oracle.sliding_window import synthetic_flair
torch.manual_seed(3)
plan = UNetPlan(patch_size=[64, 64, 64])
net = perturb_like_trained(unet_from_plan(plan), seed=4).train()
sd = state_dict_numpy(net)
vol = synthetic_flair((72, 80, 72), seed=6)[0]
data = np.stack([vol[4:68, 8:72, 4:68], vol[8:72, 16:80, 0:64][::-1].copy()])[:, None].astype(np.float32)
rng = np.random.default_rng(8)
labels = np.stack([(data[0, 0] > 1.0), rng.random((64, 64, 64)) < 0.1]).astype(np.uint8)
weights = ot.deep_supervision_weights(plan.num_pool)
x = torch.from_numpy(data)
tgt = ot.downsample_targets(torch.from_numpy(labels[:, None].astype(np.float32)), plan.num_pool)
net.do_ds = True
loss = ot.multiple_output_loss(net(x), tgt, weights); loss.backward()
net64 = copy.deepcopy(net).double(); net64.zero_grad()
t0 = time.time()
l64 = ot.multiple_output_loss(net64(x.double()), [t.double() for t in tgt], weights); l64.backward()
print('float64 autograd: %.1f s' % (time.time() - t0))
tr = UNetTrainer(plan, sd, 'cuda:0')
td = downsample_targets(torch.from_numpy(labels).cuda(), plan.num_pool)
tr.forward(x.cuda()); tr.backward(td); torch.cuda.synchronize()
g64 = dict((n, p.grad) for n, p in net64.named_parameters() if p.grad is not None)
gmax = max(float(g.abs().max()) for g in g64.values())
rows = []
for n, p in net.named_parameters():
    if n not in g64:
        continue
    r = g64[n]
    if float(r.abs().max()) < 1e-4 * gmax:
        continue
    e_hip = float((tr.g[n].cpu().double() - r).norm() / r.norm())
    e_cpu = float((p.grad.double() - r).norm() / r.norm())
    rows.append((e_hip, e_cpu, n))
rows.sort(reverse=True)
print('relative L2 error against float64: HIP trainer | torch CPU fp32')
for r in rows[:10]:
    print('  %-55s %.3g | %.3g' % (r[2], r[0], r[1]))
print('median HIP %.3g, median CPU %.3g' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
