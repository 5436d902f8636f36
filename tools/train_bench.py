"""Time one nnUNetTrainerV2 iteration (BASELINE plan, batch 2 of 128^3 patches) on the HIP kernels.
env: WMH_BATCH (2), WMH_ITERS (4), WMH_PRECISION (fp32 | f16x2), WMH_X3_SPLITK=0 (pair kernels without split-K slabs),
WMH_X3_S2_WGRAD=0 (strided weight gradients on the exact-fp32 kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.synthetic import he_init_state_dict
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
plan = baseline_plan()
B = int(os.environ.get('WMH_BATCH', '2'))
tr = UNetTrainer(plan, he_init_state_dict(plan, 0), 'cuda:0', precision=os.environ.get('WMH_PRECISION', 'fp32'))
if os.environ.get('WMH_X3_SPLITK') == '0':
    tr.x3_splitk = False
if os.environ.get('WMH_X3_S2_WGRAD') == '0':
    tr.x3_s2_wgrad = False
torch.manual_seed(0)
x = torch.randn(B, 1, 128, 128, 128, device='cuda')
lab = (torch.rand(B, 128, 128, 128, device='cuda') < 0.05).to(torch.uint8)
td = downsample_targets(lab, plan.num_pool)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(int(os.environ.get('WMH_ITERS', '4'))):
    e0 = ev(); tr.forward(x); e1 = ev(); l = tr.backward(td); e2 = ev(); tr.optimizer_step(1e-2); e3 = ev()
    torch.cuda.synchronize()
    print('iter %d: loss %.5f  fwd %.1f ms  bwd %.1f ms  opt %.2f ms  total %.1f ms' %
          (it, float(l.cpu()), e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e0.elapsed_time(e3)), flush=True)
