"""CPU oracle for the DeepWMH hot path (TEST INFRASTRUCTURE ONLY).

This package is a torch-CPU / numpy / scipy restatement of what the
`nnUNet_predict` subprocess launched by the reference does
(/root/reference/deepwmh/main/predict.py:153-156,
/root/reference/deepwmh/pipeline/DCNN_multistage.py:331-336,531-533).

PARITY UNPINNED for the nnU-Net rows (generic_unet, sliding_window, training, preprocess); the rows whose code IS in
/root/reference (postprocess, stage1, stage2's Dice) are pinned to the reference's own function bodies through
tests/golden/rows_ref.npz (tests/golden/make_golden_ref.py, tests/test_rows_ref.py).  The arithmetic of the network path lives in the un-vendored, un-pinned
nnU-Net fork `lchdl/nnUNet_for_DeepWMH` (README.md:69-72 of the reference; absent
from its setup.py:32-46), which is not present in /root/reference, not installed
and not fetchable.  The reference has no tests, golden vectors or sample weights
for this path.  What follows restates the published
MIC-DKFZ/nnUNet v1 (1.7.x) algorithm; the arithmetic primitives are the same
torch.nn / scipy.ndimage calls nnU-Net itself makes, so each layer is exact, but
the glue (tiling, Gaussian map, mirroring order, aggregation) is pinned only by
our own golden fixtures (tests/golden), not by reference outputs.

What is a GUESS about the fork (its source is not available; each is inferred from how the reference uses the result):
* `--save_softmax` writes ONLY channel 0 as `<case>_0.nii.gz`, i.e. P(background): DCNN_multistage.py:340-344 reads exactly
  that file name and :106-108 treats it as an inverted lesion probability.  Dtype float32 (:359 loads it as such).
* `--selected_cases` restricts the prediction to the listed case ids (padding spaces in the list: :327-333).
* `nnUNet_train -b N` = batches per epoch (num_batches_per_epoch), `-e` = epochs, `--save_every_epoch` writes
  `model_ep_%04d` (:196-216, 292-304, 331-336), `--custom_val_cases` replaces the validation split, `--noval` skips the final
  validation.  None of these exist in upstream MIC-DKFZ/nnUNet v1.
* Deep-supervision targets follow upstream's `DownsampleSegForDSTransform2` / `resize_segmentation(order=0)` pixel-centre
  coordinate map (oracle/training.py; round 1 used `seg[::k]`, which upstream does not).
Everything else (module order and names, He init, poly LR, Dice+CE, Gaussian map, mirror order, step rule) is upstream v1.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package.  The product (`deepwmh_amd`) never does.
"""
