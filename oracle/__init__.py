"""CPU oracle for the DeepWMH hot path (TEST INFRASTRUCTURE ONLY).

This package is a torch-CPU / numpy / scipy restatement of what the
`nnUNet_predict` subprocess launched by the reference does
(/root/reference/deepwmh/main/predict.py:153-156,
/root/reference/deepwmh/pipeline/DCNN_multistage.py:331-336,531-533).

PARITY UNPINNED.  The arithmetic of this path lives in the un-vendored, un-pinned
nnU-Net fork `lchdl/nnUNet_for_DeepWMH` (README.md:69-72 of the reference; absent
from its setup.py:32-46), which is not present in /root/reference, not installed
and not fetchable.  The reference has no tests, golden vectors or sample weights
for this path.  What follows restates the published
MIC-DKFZ/nnUNet v1 (1.7.x) algorithm; the arithmetic primitives are the same
torch.nn / scipy.ndimage calls nnU-Net itself makes, so each layer is exact, but
the glue (tiling, Gaussian map, mirroring order, aggregation) is pinned only by
our own golden fixtures (tests/golden), not by reference outputs.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package.  The product (`deepwmh_amd`) never does.
"""
