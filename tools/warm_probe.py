"""What the first case of an nnUNet_predict process pays before its first forward runs at speed: context, engine, allocation,
first forward (code objects, composed decoder images, buffers), second forward.  python3 tools/warm_probe.py on the GPU box.
Measured (round 5): imports + context 0.56 s, engine 0.01 s, first forward 1.40 s (4 x 4.3 GB of torch.empty alone: 0.36 s),
second forward 0.09 s, 23.8 GB allocated at batch 16."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
t0 = time.time()
from deepwmh_amd.network import UNetEngine
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.sliding_window import SlidingWindowPredictor
from deepwmh_amd.synthetic import he_init_state_dict
plan = baseline_plan(); sd = he_init_state_dict(plan, 0)
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t1 = time.time(); print('imports + context %.2f s' % (t1 - t0))
eng = UNetEngine(plan, sd, 'cuda:0', precision='f16x2'); torch.cuda.synchronize()
t2 = time.time(); print('engine (weights to device, transposed-conv images) %.2f s' % (t2 - t1))
# allocation alone: the sizes the forward will ask for
big = [torch.empty(int(4.3e9 / 4), device='cuda') for _ in range(4)]; torch.cuda.synchronize()
t3 = time.time(); print('4 x 4.3 GB torch.empty %.3f s' % (t3 - t2)); del big; torch.cuda.empty_cache()
t3 = time.time()
pred = SlidingWindowPredictor(eng, do_mirroring=True, batch=16)
pred.warm_up(); torch.cuda.synchronize()
t4 = time.time(); print('warm_up (first forward: allocation, composed images, code objects) %.2f s' % (t4 - t3))
pred.warm_up(); torch.cuda.synchronize()
t5 = time.time(); print('second forward %.3f s' % (t5 - t4))
print('allocated %.1f GB' % (torch.cuda.memory_allocated() / 1e9))
