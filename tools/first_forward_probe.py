"""Where the first second of a fresh process goes (one case through nnUNet_predict): engine construction, the first forward (buffer allocation, code
objects, composed weight images), the second forward.   python tools/first_forward_probe.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
import torch
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t1 = time.time()
from deepwmh_amd.network import UNetEngine
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.synthetic import he_init_state_dict
plan = baseline_plan(); sd = he_init_state_dict(plan, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t2 = time.time()
eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2'); torch.cuda.synchronize()
t3 = time.time()
x = torch.zeros(B, 1, 128, 128, 128, device='cuda')
import deepwmh_amd.network as nw
orig = eng._up_image
tt = [0.0]
def timed_up(st):
    a = time.time(); r = orig(st); tt[0] += time.time() - a; return r
eng._up_image = timed_up
eng.forward_features(x); torch.cuda.synchronize()
t4 = time.time()
eng.forward_features(x); torch.cuda.synchronize()
t5 = time.time()
print('import torch + context %.2f s | weights (He init on the host) %.2f | engine construction %.2f | first forward of %d patches %.2f (of it composing the '
      'decoder weight images on the host %.2f) | second forward %.3f' % (t1 - t0, t2 - t1, t3 - t2, B, t4 - t3, tt[0], t5 - t4))
