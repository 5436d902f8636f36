"""Stage I (NLL initial segmentation) of one BASELINE-sized case (160x192x160, R references): wall time of
deepwmh_amd.stage1.nll_analysis_arrays on the GPU (host arrays in, host arrays out) and of the numpy oracle on the host.
usage: python tools/stage1_bench.py [R] [--no-cpu]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from stage1_cases import make_case                            # noqa: E402
from deepwmh_amd import stage1                                 # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
shape = (160, 192, 160)
x, refs, l1, l2 = make_case(shape, R, 0)
vs = [1.0, 1.0, 1.0]
stage1.nll_analysis_arrays(x, refs, l1, l2, vs, '+')            # warm-up (allocator, code objects)
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    out = stage1.nll_analysis_arrays(x, refs, l1, l2, vs, '+')
torch.cuda.synchronize()
gpu = (time.time() - t) / 3
print('GPU  nll_analysis_arrays: %.3f s per case (R = %d, %s), threshold %.3f, %d lesion voxels'
      % (gpu, R, 'x'.join(map(str, shape)), out['threshold'], int(out['segmentation'].sum())))
if '--no-cpu' not in sys.argv:
    from oracle import stage1 as o
    t = time.time()
    ref = o.nll_analysis_arrays(x, refs, l1, l2, vs, '+')
    cpu = time.time() - t
    print('CPU  oracle (numpy/scipy, 1 process): %.1f s per case -> x%.0f; |threshold diff| %.2e, segmentation mismatches %d'
          % (cpu, cpu / gpu, abs(ref['threshold'] - out['threshold']), int((ref['segmentation'] != out['segmentation']).sum())))
