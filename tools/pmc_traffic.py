"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
a calibration launch of known byte count in the conv kernels' own access width (4 B per lane),
then two forwards of the BASELINE network at WMH_BATCH patches (tools/profile_round.sh: 16, the bench's batch).  tools/parse_pmc.py reads the CSVs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from deepwmh_amd import _lib  # noqa: E402
from deepwmh_amd.network import UNetEngine  # noqa: E402
from deepwmh_amd.plan import baseline_plan  # noqa: E402
from deepwmh_amd.synthetic import he_init_state_dict  # noqa: E402

plan = baseline_plan()
MODE = os.environ.get('WMH_PMC_MODE', 'fp32')           # fp32 | bf16 (bf16 operands, fp32 tensors) | bf16_storage (channels-last bf16) | f16x2
eng = UNetEngine(plan, he_init_state_dict(plan, 0), 'cuda:0', precision=MODE if MODE in ('fp32', 'f16x2') else 'bf16',
                 storage='bf16' if MODE == 'bf16_storage' else 'fp32')
lib = eng.lib
n = 256 * 1024 * 1024                      # 1 GiB read + 1 GiB written, far beyond the 256 MiB Infinity Cache
a = torch.randn(n, device='cuda')
b = torch.empty_like(a)
ss = torch.tensor([[1.5, 0.25]], device='cuda')
for _ in range(2):
    _lib.check(lib.wmh_instnorm_lrelu_apply(_lib.ptr(a), _lib.ptr(ss), _lib.ptr(b), 1, 1, n, 0.01, None), 'apply')
torch.cuda.synchronize()
del a, b
B = int(os.environ.get('WMH_BATCH', '16'))
x = torch.randn(B, 1, 128, 128, 128, device='cuda')
for _ in range(2):
    eng.forward_features(x)
torch.cuda.synchronize()
print('done')
