"""Diagnostic: full-size BASELINE forward in storage='bf16', then report non-finite values per engine buffer."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd.network import UNetEngine
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
plan = baseline_plan()
sd = he_init_state_dict(plan, seed=0)
eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16', storage='bf16')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vol = synthetic_flair((160, 192, 160), seed=0)
x = torch.from_numpy(np.stack([vol[:, :128, :128, :128]] * N)).cuda().contiguous()
outs = eng.forward_features(x)
torch.cuda.synchronize()
for k in sorted(eng._bufs):
    t = eng._bufs[k]
    f = t.float()
    bad = int((~torch.isfinite(f)).sum())
    print('%-16s %-10s n=%-11d nonfinite=%-9d absmax=%.4g' % (k, str(t.dtype).replace('torch.', ''), t.numel(), bad, float(f[torch.isfinite(f)].abs().max()) if bad < t.numel() else float('nan')))
